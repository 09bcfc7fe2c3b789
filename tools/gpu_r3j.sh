#!/bin/bash
tag=${1:-r3j}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "head or temporal or tgemm" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/tattn_time.py > $O/stats.log 2>&1
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("tattn","tgemm","head_fwd")): print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:6.1f} us min {float(r['MinNs'])/1e3:6.1f}")
PY
cd $R
B="python bench.py --no-cpu-baseline --sustain-seconds 0 --steps 20 --warmup 3"
for rep in 1 2; do $B > $O/bench_$rep.json 2> $O/bench_$rep.err; echo "$(head -c 200 $O/bench_$rep.json | tail -c 60)"; done
