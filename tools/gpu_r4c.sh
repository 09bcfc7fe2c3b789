#!/bin/bash
# round 4: (1) fused MLP after the latency fixes: tests + stand-alone A/B; (2) CLS-only last block: kernel tests, equivalence
# with the full computation, model / bench-size parity vs the oracle; (3) the step with and without the pruning
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4c; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "mlp_fused or cls_query or gemm_ln" > $O/pytest_kernels.log 2>&1; tail -4 $O/pytest_kernels.log
timeout 300 python tools/one_mlp.py 8 > $O/one_mlp.log 2>&1; cat $O/one_mlp.log
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x > $O/pytest_model.log 2>&1; tail -4 $O/pytest_model.log
for pr in 1 0; do
  SAIS_VIT_PRUNE_LAST=$pr timeout 600 python bench.py --steps 30 --warmup 5 --sustain-seconds 0 --no-cpu-baseline > $O/bench_prune$pr.json 2> $O/bench_prune$pr.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r4c/bench_prune$pr.json"))
print("prune=$pr", d["value"], d["ms_per_step"], d["parity"], {k: (v["avg_us"], v["launches_per_step"]) for k, v in d["roofline"]["all_kernels"].items()})
PY
done
timeout 1200 python -m pytest tests/test_bench_size_gpu.py tests/test_dropout_gpu.py tests/test_inference_gpu.py -m gpu -q -x > $O/pytest_size.log 2>&1; tail -4 $O/pytest_size.log
