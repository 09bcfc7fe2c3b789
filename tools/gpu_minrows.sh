#!/bin/bash
# A/B: switch-over of the row-owning GEMM between "few rows per workgroup, two workgroups per CU" and "whole 112-row tiles"
# (SAIS_ROW_MINROWS; default 64) at mid-size M: 4-clip training step (M = 25 216), 2-clip (12 608), the DINO step (teacher 25 216)
tag=${1:-minrows}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
for rep in 1 2; do
for mr in 0 48 24 12; do
  for clips in 4 2; do
    SAIS_ROW_MINROWS=$mr python bench.py --clips $clips --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 > $O/b_${mr}_${clips}_$rep.json 2> $O/b_${mr}_${clips}_$rep.err
    echo "minrows $mr clips $clips rep $rep $(grep -o '"ms_per_step": [0-9.]*' $O/b_${mr}_${clips}_$rep.json | head -1)"
  done
  SAIS_ROW_MINROWS=$mr python bench.py --workload dino --steps 10 --warmup 3 --no-cpu-baseline > $O/d_${mr}_$rep.json 2> $O/d_${mr}_$rep.err
  echo "minrows $mr dino rep $rep $(grep -o '"ms_per_step": [0-9.]*' $O/d_${mr}_$rep.json | head -1)"
done
done
