#!/bin/bash
tag=${1:-r3g}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
B="python bench.py --no-cpu-baseline --sustain-seconds 0 --steps 20 --warmup 3"
run() {
  env $1 $B > $O/b.json 2> $O/b.err
  python - $O/b.json "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); ak=d["roofline"]["all_kernels"]
g=lambda k: ak[k]["avg_us"]
print(f"{sys.argv[2]:36s} {d['ms_per_step']:.3f}  fc1 {g('gemm_nt<gelu_grad_bf16>[N1536,K384]')} dxfc2 {g('gemm_nt<mul_bf16>[N1536,K384]')} qkv {g('gemm_nt<bias_bf16>[N1152,K384]')} afwd {g('vit_attn_fwd')} abwd {g('vit_attn_bwd')}")
PY
}
for rep in 1 2; do
  run "SAIS_W8P_PRIO=0 SAIS_ATTN_PRIO=0"
  run "SAIS_W8P_PRIO=1 SAIS_ATTN_PRIO=1"
  run "SAIS_W8P_PRIO=2 SAIS_ATTN_PRIO=2"
  run "SAIS_W8P_PRIO=3 SAIS_ATTN_PRIO=4"
  run "SAIS_W8P_PRIO=2 SAIS_ATTN_PRIO=3"
done
