#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_dino_gpu.py -m gpu -q -k "attention or attn or multigroup or backbone" 2>&1 | tail -3
for rep in 1 2 3; do python tools/attn_time.py 2>&1 | tail -1; done
for rep in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 | grep -o '"ms_per_step": [0-9.]*' | head -1; done
python bench.py --workload dino --steps 10 --warmup 3 --no-cpu-baseline | grep -o '"ms_per_step": [0-9.]*'
