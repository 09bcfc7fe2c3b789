#!/bin/bash
# A/B: higher issue priority for the younger sibling wave (w + 4) of each SIMD in the persistent NT GEMM (-DSAIS_NT_SIBPRIO)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
SAIS_HIP_LIB=$R/tools/bin/sibstamp/libsais_hip.so python tools/nt_stamp.py 1536 384 10 | cut -c1-60,215-330
for rep in 1 2; do
for v in base sibprio; do
  lib=$R/sais_amd/libsais_hip.so; [ $v != base ] && lib=$R/tools/bin/$v/libsais_hip.so
  for shape in "1536 384 10" "1536 384 11" "1152 384 0"; do echo "$v [$shape] $(SAIS_HIP_LIB=$lib python tools/one_gemm.py $shape 20 2>&1 | tail -1)"; done
  echo "$v $(SAIS_HIP_LIB=$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 | grep -o '"ms_per_step": [0-9.]*' | head -1)"
done
done
