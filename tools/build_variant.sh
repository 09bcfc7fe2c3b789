#!/bin/bash
# Build a variant of libsais_hip.so with extra compiler flags into tools/bin/<name>/ (git-ignored; travels with gpurun),
# for A/B runs inside the training step:  SAIS_HIP_LIB=tools/bin/<name>/libsais_hip.so python bench.py ...
#   tools/build_variant.sh norow -DSAIS_NO_ROW_PLAIN
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/tools/bin/$name
mkdir -p $out
for f in gemm gemm_row norm attn_vit misc temporal tgemm tattn preprocess dino; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-result \
      -mllvm -amdgpu-mfma-vgpr-form=1 "$@" -c $root/sais_amd/csrc/$f.hip -o $out/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $out/*.o -o $out/libsais_hip.so
echo built $out/libsais_hip.so
