#!/bin/bash
# The driver's own command (python bench.py, default flags) with its wall time; prints the headline and the variants
out=gpurun_out/${1:-bench_default}
mkdir -p $out
t0=$(date +%s)
python bench.py > $out/bench.json 2> $out/bench.err
echo "rc=$? wall=$(( $(date +%s) - t0 )) s"
python - <<PY
import json
d = json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(d["value"], d["ms_per_step"], r["kernel"], r["frac"], r["avg_launch_us"], r["traffic"], r.get("traffic_stale"))
print(json.dumps(d.get("variants"), indent=1)[:3500])
print(d["sustained"]); print(d["cpu_baseline"]["value"], d["parity"])
PY
