#!/bin/bash
# DINO objective bring-up on the GPU box: the new tests, then the kernel tests that share the re-templated attention.
tag=${1:-dino_a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_dino_gpu.py -m gpu -q -x --timeout 900 > $O/pytest_dino.log 2>&1; tail -40 $O/pytest_dino.log
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x > $O/pytest_kernels.log 2>&1; tail -5 $O/pytest_kernels.log
