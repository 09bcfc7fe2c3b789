#!/bin/bash
# A/B: 112-B rows for the dS image of the attention backward (conflict-free for the 8-B writes as well as the transposed reads)
tag=${1:-srow}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
for rep in 1 2; do
for v in base srow112; do
  lib=$R/sais_amd/libsais_hip.so; [ $v != base ] && lib=$R/sais_amd/libsais_hip_$v.so
  SAIS_HIP_LIB=$lib python tools/attn_time.py > $O/attn_${v}_$rep.txt 2>&1; echo "$v $rep $(tail -1 $O/attn_${v}_$rep.txt)"
  SAIS_HIP_LIB=$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
  echo "$v $rep $(grep -o '"ms_per_step": [0-9.]*' $O/bench_${v}_$rep.json | head -1)"
done
done
SAIS_HIP_LIB=$R/sais_amd/libsais_hip_srow112.so timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k attention 2>&1 | tail -2
