#!/bin/bash
# round 4: CLS attention kernels (row-group layout), pruned step, whole GPU suite
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4d; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "cls_query" > $O/pytest_cls.log 2>&1; tail -3 $O/pytest_cls.log
timeout 600 python bench.py --steps 30 --warmup 5 --sustain-seconds 0 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d = json.load(open("gpurun_out/r4d/bench.json"))
print(d["value"], d["ms_per_step"], d["step_tflops"], d["frac_of_mfma_roofline"], d["parity"]["max_abs_logit"])
print({k: (v["avg_us"], v["launches_per_step"]) for k, v in d["roofline"]["all_kernels"].items() if "cls" in k or "K1536]" in k or "tn" in k})
PY
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -6 $O/pytest.log
cp gpurun_out/parity_worst.json $O/parity_worst.json 2>/dev/null
