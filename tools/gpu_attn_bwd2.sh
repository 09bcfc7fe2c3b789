#!/bin/bash
# A/B: attention backward with two key tiles per wave (512 threads, SAIS_ATTN_BWD2=1) vs the 16-wave kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
SAIS_ATTN_BWD2=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "attention" 2>&1 | tail -2
for rep in 1 2; do
  echo "base $(python tools/attn_time.py 2>&1 | tail -1)"
  echo "bwd2 $(SAIS_ATTN_BWD2=1 python tools/attn_time.py 2>&1 | tail -1)"
done
for rep in 1 2; do
  echo "base $(python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 | grep -o '"ms_per_step": [0-9.]*' | head -1)"
  echo "bwd2 $(SAIS_ATTN_BWD2=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 | grep -o '"ms_per_step": [0-9.]*' | head -1)"
done
