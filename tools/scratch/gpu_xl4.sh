#!/bin/bash
out=gpurun_out/${1:-r6h}
mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "slab_mode or gemm_tn_grouped_matches" > $out/test.log 2>&1
echo "pytest rc=$?" >> $out/summary.txt; tail -1 $out/test.log >> $out/summary.txt
for rep in 1 2; do
  for spec in "SAIS_TN_XL=0" "SAIS_TN_SLABS=1" "SAIS_TN_XL_SLABS=1"; do
    echo -n "$spec rep=$rep: " >> $out/summary.txt
    env $spec timeout 120 python tools/tn_only.py 20 2>/dev/null >> $out/summary.txt
  done
done
cat $out/summary.txt
bash tools/gpu_step_ab.sh ${1:-r6h}_step SAIS_TN_XL=0 SAIS_TN_SLABS=1 SAIS_TN_XL_SLABS=1 | grep -o "^SAIS.*rep [12]: [0-9.]* [0-9.]*\|gemm_tn_grouped\[4 GEMMs,M50432\]=[0-9.]*" | paste - -
