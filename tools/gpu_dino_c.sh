#!/bin/bash
tag=${1:-dino_c}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_dino_gpu.py -m gpu -q -x --timeout 900 -k "two_ranks or cli" > $O/pytest_dino.log 2>&1; tail -40 $O/pytest_dino.log
