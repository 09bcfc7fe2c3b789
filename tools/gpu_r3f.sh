#!/bin/bash
tag=${1:-r3f}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
B="python bench.py --no-cpu-baseline --sustain-seconds 0 --steps 20 --warmup 3"
for rep in 1 2; do
  for cfg in "0 0" "1 0" "1 2" "0 2"; do
    set -- $cfg
    SAIS_ATTN_QUEUE=$1 SAIS_ATTN_STAGGER=$2 $B > $O/bench_q$1_s$2_$rep.json 2> $O/bench_q$1_s$2_$rep.err
    python - $O/bench_q$1_s$2_$rep.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); ak=d["roofline"]["all_kernels"]
print(sys.argv[1].split("/")[-1], d["ms_per_step"], "attn_bwd", ak["vit_attn_bwd"]["avg_us"], "dW", ak["gemm_tn_grouped"]["avg_us"])
PY
  done
done
