#!/bin/bash
# rocprofv3 kernel durations of the stand-alone dW launch pair
out=gpurun_out/${1:-r6n}; mkdir -p $out
python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm_tn_grouped" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/st -- python3 $GRAFT_REPO_ROOT/tools/tn_only.py 30 > $GRAFT_REPO_ROOT/$out/log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv,glob
f=glob.glob("$out/st/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "xl" in r["Name"] or "tn_" in r["Name"]: print(r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3)
PY
find $out -name "*kernel_trace.csv" -delete
