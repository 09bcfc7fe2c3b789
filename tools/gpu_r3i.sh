#!/bin/bash
tag=${1:-r3i}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "temporal or tn or tgemm" > $O/pytest_k.log 2>&1; tail -4 $O/pytest_k.log
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_dropout_gpu.py tests/test_bench_size_gpu.py tests/test_train_gpu.py -m gpu -q > $O/pytest_m.log 2>&1; tail -4 $O/pytest_m.log
B="python bench.py --no-cpu-baseline --sustain-seconds 0 --steps 20 --warmup 3"
for rep in 1 2; do $B > $O/bench_$rep.json 2> $O/bench_$rep.err; echo "$(head -c 200 $O/bench_$rep.json | tail -c 60)"; done
