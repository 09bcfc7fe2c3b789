#!/bin/bash
# compile gemm_tn_xl.hip with resource remarks and split the per-kernel assembly into /tmp/xl4.s, /tmp/xl8.s
cd /root/repo/sais_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-result \
  -Rpass-analysis=kernel-resource-usage -save-temps=obj "$@" -c gemm_tn_xl.hip -o /tmp/xl.o 2>&1 | grep -E "error|warning|Name|VGPRs:|AGPRs|Scratch|Spill" 
S=/tmp/gemm_tn_xl-hip-amdgcn-amd-amdhsa-gfx950.s
awk '/^_ZN12_GLOBAL__N_117gemm_tn_xl_kernelILi4/,/s_endpgm/' $S > /tmp/xl4.s
awk '/^_ZN12_GLOBAL__N_117gemm_tn_xl_kernelILi8/,/s_endpgm/' $S > /tmp/xl8.s
wc -l /tmp/xl4.s /tmp/xl8.s
