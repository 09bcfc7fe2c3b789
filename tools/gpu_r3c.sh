#!/bin/bash
# round-3: row-owning GEMM with role-split staging (A stream two K-steps ahead) — parity, then A/B inside the step
tag=${1:-r3c}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm_ln or gemm_nt_epilogues" > $O/pytest_kernels.log 2>&1; tail -3 $O/pytest_kernels.log
B="python bench.py --no-cpu-baseline --sustain-seconds 0 --steps 20 --warmup 3"
for rep in 1 2; do
  SAIS_ROW_WAVES=4 $B > $O/bench_w4_$rep.json 2> $O/bench_w4_$rep.err; head -c 230 $O/bench_w4_$rep.json | tail -c 120; echo
  SAIS_ROW_STAG=0 SAIS_ROW_SPEC=1 $B > $O/bench_spec_$rep.json 2> $O/bench_spec_$rep.err; head -c 230 $O/bench_spec_$rep.json | tail -c 120; echo
  SAIS_ROW_STAG=1 SAIS_ROW_SPEC=1 $B > $O/bench_specstag_$rep.json 2> $O/bench_specstag_$rep.err; head -c 230 $O/bench_specstag_$rep.json | tail -c 120; echo
done
timeout 1200 python -m pytest tests/test_bench_size_gpu.py tests/test_dropout_gpu.py -m gpu -q -x > $O/pytest_step.log 2>&1; tail -3 $O/pytest_step.log
