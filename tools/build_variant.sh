#!/bin/bash
# Build a variant of libsais_hip.so with extra compiler flags into tools/bin/<name>/ (git-ignored; travels with gpurun),
# for A/B runs inside the training step:  SAIS_HIP_LIB=tools/bin/<name>/libsais_hip.so python bench.py ...
#   tools/build_variant.sh clk -DSAIS_CLK_STAMP
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/tools/bin/$name
mkdir -p $out
srcs=$(sed -n 's/^SRCS = //p' $root/sais_amd/csrc/Makefile)
for s in $srcs; do
  f=${s%.hip}
  extra="-mllvm -amdgpu-mfma-vgpr-form=1"
  [ "$f" = mlp_fused ] && extra=""
  [ "$f" = gemm_tn_xl ] && extra=""
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-result \
      $extra "$@" -c $root/sais_amd/csrc/$f.hip -o $out/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $out/*.o -o $out/libsais_hip.so
rm -f $out/*.o
echo built $out/libsais_hip.so
