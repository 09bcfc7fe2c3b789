#!/bin/bash
# round-3 first GPU call: validate the new bench paths (graph-captured DP step, two-stream, DropPath cost, sustained rate)
tag=${1:-r3a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
timeout 300 python bench.py --no-cpu-baseline > $O/bench_dp01.json 2> $O/bench_dp01.err; tail -c 900 $O/bench_dp01.json | head -c 500; echo
timeout 300 python bench.py --no-cpu-baseline --vit-drop-path 0 --sustain-seconds 0 > $O/bench_dp0.json 2> $O/bench_dp0.err; head -c 300 $O/bench_dp0.json; echo
timeout 300 python bench.py --no-cpu-baseline --two-stream --sustain-seconds 0 > $O/bench_two.json 2> $O/bench_two.err; head -c 300 $O/bench_two.json; echo; tail -3 $O/bench_two.err
SAIS_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --sustain-seconds 0 > $O/force_dist_graph.log 2>&1; tail -c 1500 $O/force_dist_graph.log | head -c 700; echo
SAIS_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 \
    bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --sustain-seconds 0 --no-graph > $O/force_dist_eager.log 2>&1; tail -c 1500 $O/force_dist_eager.log | head -c 300; echo
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
