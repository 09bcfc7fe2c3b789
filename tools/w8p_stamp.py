#!/usr/bin/env python3
"""Where a wave of the persistent NT kernel spends its time (s_memtime stamps per phase).  Needs an instrumented build:
    make -C sais_amd/csrc clean && make -C sais_amd/csrc EXTRA="-mllvm -amdgpu-mfma-vgpr-form=1 -DSAIS_W8P_STAMP"
(then rebuild normally)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from sais_amd import _lib as L, ops
M = 50432
for name, N, K in (("qkv", 1152, 384), ("dXfc1", 384, 1536)):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    dbg = torch.zeros(512 * 8 * 8, dtype=torch.int64, device="cuda")
    for _ in range(3):
        ops.gemm_nt(a, w, L.EPI_BIAS_BF16, out, bias=b, aux=dbg.view(torch.bfloat16).view(-1, 8), grp=(0, 777, 0))
    torch.cuda.synchronize()
    d = dbg.view(512, 8, 8).float()
    tot = d[:, :, 6].mean().item()
    names = ["issue", "mma", "vmwait", "barrier", "epilogue", "acc-init/other"]
    print(name, "cycles per WG-wave total %.0f:" % tot, {n: "%.0f%%" % (100 * d[:, :, i].mean().item() / tot) for i, n in enumerate(names)})
