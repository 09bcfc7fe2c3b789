#!/bin/bash
# A/B: temporal attention with all staging loads in flight + the key flags in LDS (vs the previous library in tools/bin/oldtattn)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_dropout_gpu.py -m gpu -q -x 2>&1 | tail -2
for rep in 1 2; do
for v in old new; do
  lib=$R/sais_amd/libsais_hip.so; [ $v = old ] && lib=$R/tools/bin/oldtattn/libsais_hip.so
  echo "== $v $rep"; SAIS_HIP_LIB=$lib python tools/tattn_time.py 2>&1 | grep -i "fwd\|bwd" | head -6
  SAIS_HIP_LIB=$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 | grep -o '"ms_per_step": [0-9.]*' | head -1
done
done
