#!/bin/bash
# round 4, first call: the new parity tests + the bench line with the in-line parity gate
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4a; mkdir -p $O; cd $R
python -m pytest tests/test_bench_size_gpu.py -m gpu -q -x > $O/pytest_bench_size.log 2>&1; tail -5 $O/pytest_bench_size.log
cp gpurun_out/parity_worst.json $O/parity_worst.json 2>/dev/null
python bench.py --steps 30 --warmup 5 --sustain-seconds 5 > $O/bench.json 2> $O/bench.err; head -c 600 $O/bench.json; echo; tail -3 $O/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4a/bench.json"))
print("parity", d.get("parity"))
print({k: v["avg_us"] for k, v in d["roofline"]["all_kernels"].items()})
PY
python -m pytest tests -m gpu -q -x --deselect tests/test_bench_size_gpu.py > $O/pytest.log 2>&1; tail -3 $O/pytest.log
