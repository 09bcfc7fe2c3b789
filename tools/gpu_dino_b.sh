#!/bin/bash
tag=${1:-dino_b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
python bench.py --workload dino --steps 5 --warmup 2 > $O/dino_bench.json 2> $O/dino_bench.err; cat $O/dino_bench.json; tail -3 $O/dino_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --workload dino --steps 5 --warmup 2 --no-cpu-baseline > $O/stats.log 2>&1
find $O -name "*kernel_trace.csv" -size +20M -delete
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); head -40 $f | cut -c1-200
