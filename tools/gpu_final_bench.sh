#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3final}
mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.err; head -c 300 $O/bench.json; echo
python - $O/bench.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(r["kernel"], r["achieved"], r["frac"], r["avg_launch_us"], r["traffic"], r.get("hbm_frac_counters"), d["sustained"])
PY
