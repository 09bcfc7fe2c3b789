#!/bin/bash
# Variant of libsais_hip.so in which ONE source is rebuilt with extra flags (the other objects are the main build's):
#   tools/build_variant_file.sh <name> <source stem> [flags...]   ->  tools/bin/<name>/libsais_hip.so  (SAIS_HIP_LIB=...)
set -e
name=$1; stem=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/tools/bin/$name
mkdir -p $out
make -s -C $root/sais_amd/csrc -j8 >/dev/null
cp $root/sais_amd/csrc/*.o $out/
extra="-mllvm -amdgpu-mfma-vgpr-form=1"
[ "$stem" = mlp_fused ] && extra=""
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-result $extra "$@" \
    -c $root/sais_amd/csrc/$stem.hip -o $out/$stem.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $out/*.o -o $out/libsais_hip.so
rm -f $out/*.o
echo built $out/libsais_hip.so
