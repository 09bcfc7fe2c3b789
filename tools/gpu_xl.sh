#!/bin/bash
# A/B of the dW kernels on one box: parity tests with the 192x384 tile (4 / 8 waves), then stand-alone timings + ablations
out=gpurun_out/${1:-r6b}
mkdir -p $out
for w in 4 8; do
  SAIS_TN_XL=$w timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm_tn_grouped_matches or gemm_tn_exact or test_gemm_tn" > $out/test_xl$w.log 2>&1
  echo "XL=$w pytest rc=$?" >> $out/summary.txt; tail -1 $out/test_xl$w.log >> $out/summary.txt
done
for rep in 1 2; do
  for w in 0 4 8; do
    echo -n "XL=$w rep=$rep: " >> $out/summary.txt
    SAIS_TN_XL=$w timeout 120 python tools/tn_only.py 20 2>/dev/null >> $out/summary.txt
  done
done
for a in 1 2 4 8 3 7 15; do
  for w in 4 8; do
    echo -n "ABL=$a XL=$w: " >> $out/summary.txt
    SAIS_HIP_LIB=tools/bin/xlabl$a/libsais_hip.so SAIS_TN_XL=$w timeout 120 python tools/tn_only.py 20 2>/dev/null >> $out/summary.txt
  done
done
cat $out/summary.txt
