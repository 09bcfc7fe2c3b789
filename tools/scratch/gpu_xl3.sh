#!/bin/bash
out=gpurun_out/${1:-r6f}
mkdir -p $out
for sl in 1 0; do
  SAIS_TN_XL_SLABS=$sl timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm_tn_grouped_matches or gemm_tn_exact or test_gemm_tn" > $out/test_sl$sl.log 2>&1
  echo "SLABS=$sl pytest rc=$?" >> $out/summary.txt; tail -1 $out/test_sl$sl.log >> $out/summary.txt
done
SAIS_TN_XL=8 timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm_tn_grouped_matches" > $out/test_w8.log 2>&1
echo "XL=8 slabs pytest rc=$?" >> $out/summary.txt; tail -1 $out/test_w8.log >> $out/summary.txt
for rep in 1 2; do
  for spec in "SAIS_TN_XL=0" "SAIS_TN_XL_SLABS=0" "SAIS_TN_XL_SLABS=1" "SAIS_TN_XL=8"; do
    echo -n "$spec rep=$rep: " >> $out/summary.txt
    env $spec timeout 120 python tools/tn_only.py 20 2>/dev/null >> $out/summary.txt
  done
done
cat $out/summary.txt
bash tools/gpu_step_ab.sh ${1:-r6f}_step SAIS_TN_XL=0 SAIS_TN_XL_SLABS=0 SAIS_TN_XL_SLABS=1 | grep -o "^SAIS.*rep [12]: [0-9.]* [0-9.]*\|gemm_tn_grouped\[4 GEMMs,M50432\]=[0-9.]*" | paste - -
