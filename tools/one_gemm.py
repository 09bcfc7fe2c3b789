#!/usr/bin/env python3
"""Run one GEMM shape a few times (for rocprofv3 --pmc passes).  usage: one_gemm.py N K epilogue [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sais_amd import ops  # noqa: E402

M = 50432
N, K, epi = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
bias = torch.randn(N, device="cuda")
f32 = epi in (2, 3)
out = torch.empty(M, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
aux = torch.randn(M, N, device="cuda") if epi == 3 else (torch.randn(M, N, device="cuda").bfloat16() if epi in (5, 6) else None)
out2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if epi == 4 else None
for _ in range(reps):
    ops.gemm_nt(a, w, epi, out, bias=bias, out2=out2, aux=aux)
torch.cuda.synchronize()
