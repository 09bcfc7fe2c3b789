#!/bin/bash
# A/B on one box: non-temporal stores in the NT GEMM epilogues (SAIS_NT_STORE = 0 / 1 / 2 builds of the library)
tag=${1:-nt}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
for rep in 1 2; do
for v in base nt1 nt2; do
  lib=$R/sais_amd/libsais_hip.so; [ $v != base ] && lib=$R/sais_amd/libsais_hip_$v.so
  SAIS_HIP_LIB=$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
  echo "$v $rep $(grep -o '"ms_per_step": [0-9.]*' $O/bench_${v}_$rep.json)"
done
done
