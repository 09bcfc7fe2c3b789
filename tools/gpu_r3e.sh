#!/bin/bash
# round-3: attention backward with the device-wide problem queue and staggered starts — parity, then A/B inside the step
tag=${1:-r3e}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "attention or multidomain or mil or vit_grads" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
B="python bench.py --no-cpu-baseline --sustain-seconds 0 --steps 20 --warmup 3"
for rep in 1 2; do
  for st in 0 1 2 4; do
    SAIS_ATTN_STAGGER=$st $B > $O/bench_st${st}_$rep.json 2> $O/bench_st${st}_$rep.err; echo "stagger $st: $(head -c 230 $O/bench_st${st}_$rep.json | tail -c 100)"
  done
done
