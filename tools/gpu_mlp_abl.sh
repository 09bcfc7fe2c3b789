#!/bin/bash
# timing ablations of the fused MLP kernel (SAIS_MLP_ABL builds: wrong results by construction, stand-alone launches)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/mlp_abl; mkdir -p $O; cd $R
for v in base "$@"; do
  lib=$R/sais_amd/libsais_hip.so; [ $v != base ] && lib=$R/tools/bin/$v/libsais_hip.so
  echo "== $v"; SAIS_HIP_LIB=$lib timeout 200 python tools/one_mlp.py 6 2>/dev/null | grep fused
done | tee $O/abl.log
