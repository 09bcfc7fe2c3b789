#!/bin/bash
# round-3: temporal encoder on sais_tgemm + slab-consuming row kernels — kernel tests, model parity, bench
tag=${1:-r3d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "tgemm or temporal" > $O/pytest_kernels.log 2>&1; tail -5 $O/pytest_kernels.log
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_dropout_gpu.py tests/test_train_gpu.py -m gpu -q > $O/pytest_model.log 2>&1; tail -5 $O/pytest_model.log
python bench.py --no-cpu-baseline --sustain-seconds 0 --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; head -c 300 $O/bench.json; echo; tail -3 $O/bench.err
