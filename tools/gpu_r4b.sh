#!/bin/bash
# round 4: fused MLP kernels: correctness at the kernel level, stand-alone timing A/B, then the whole step both ways
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4b; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "mlp_fused" > $O/pytest_mlp.log 2>&1; tail -5 $O/pytest_mlp.log
timeout 300 python tools/one_mlp.py 10 > $O/one_mlp.log 2>&1; cat $O/one_mlp.log
for f in 1 0; do
  SAIS_MLP_FUSED=$f timeout 600 python bench.py --steps 30 --warmup 5 --sustain-seconds 0 --no-cpu-baseline > $O/bench_fused$f.json 2> $O/bench_fused$f.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r4b/bench_fused$f.json"))
print("fused=$f", d["value"], d["ms_per_step"], d["parity"]["max_abs_logit"], {k: v["avg_us"] for k, v in d["roofline"]["all_kernels"].items() if "mlp" in k or "gelu" in k or "K1536" in k or "mul" in k})
PY
done
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_bench_size_gpu.py -m gpu -q -x > $O/pytest_model.log 2>&1; tail -3 $O/pytest_model.log
