#!/bin/bash
# Variant of libsais_hip.so in which only gemm_tn_xl.hip is rebuilt with extra flags (the other objects are the in-tree ones):
#   tools/build_xl_variant.sh abl1 -DSAIS_XL_ABL=1   ->  tools/bin/abl1/libsais_hip.so   (A/B through SAIS_HIP_LIB)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/tools/bin/$name
mkdir -p $out
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-result "$@" \
    -c $root/sais_amd/csrc/gemm_tn_xl.hip -o $out/gemm_tn_xl.o
objs=$(ls $root/sais_amd/csrc/*.o | grep -v gemm_tn_xl.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $out/gemm_tn_xl.o -o $out/libsais_hip.so
rm -f $out/gemm_tn_xl.o
echo built $out/libsais_hip.so
