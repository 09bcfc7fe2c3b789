#!/bin/bash
# barrier-free attention backward vs the single-pass kernel: parity tests, stand-alone timing, step A/B on one box
out=gpurun_out/${1:-r6o}; mkdir -p $out
for nb in 1 0; do
  SAIS_ATTN_BWD_NB=$nb timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "vit_attention" 2>&1 | tail -1
done
for rep in 1 2; do for nb in 0 1; do echo -n "NB=$nb: "; SAIS_ATTN_BWD_NB=$nb timeout 100 python tools/attn_time.py 2>/dev/null; done; done
bash tools/gpu_step_ab.sh ${1:-r6o}_step SAIS_ATTN_BWD_NB=0 SAIS_ATTN_BWD_NB=1 | grep -o "^SAIS.*rep [12]: [0-9.]* [0-9.]*\|vit_attn_bwd=[0-9.]*" | paste - -
