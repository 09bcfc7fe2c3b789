#!/bin/bash
# Timing ablations of the attention backward (libraries built with -DSAIS_ATTN_ABL=<mask>, tools/build_variant.sh abl<mask>):
# which phase of the query step paces the kernel?  Results of the ablated kernels are wrong by construction; only time counts.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
for a in base 1 2 4 8 32 6 14 16; do
  lib=$R/sais_amd/libsais_hip.so; [ $a != base ] && lib=$R/tools/bin/abl$a/libsais_hip.so
  echo "abl $a rep $rep: $(SAIS_HIP_LIB=$lib python tools/attn_time.py 2>&1 | tail -1)"
done
done
