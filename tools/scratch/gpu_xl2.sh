#!/bin/bash
out=gpurun_out/${1:-r6d}
mkdir -p $out
for a in 13 9 14 10 12 8; do
  for w in 4 8; do
    echo -n "ABL=$a XL=$w: " >> $out/summary.txt
    SAIS_HIP_LIB=tools/bin/xlabl$a/libsais_hip.so SAIS_TN_XL=$w timeout 120 python tools/tn_only.py 20 2>/dev/null >> $out/summary.txt
  done
done
cat $out/summary.txt
