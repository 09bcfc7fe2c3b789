#!/bin/bash
tag=${1:-dino_f}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_dino_gpu.py -m gpu -q --timeout 900 > $O/pytest_dino.log 2>&1; tail -15 $O/pytest_dino.log
python bench.py --workload dino --steps 10 --warmup 3 > $O/bench_dino.json 2> $O/bench_dino.err; head -c 600 $O/bench_dino.json; echo; tail -2 $O/bench_dino.err
