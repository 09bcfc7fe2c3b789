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
aux = torch.randn(M, N, device="cuda") if epi == 3 else (torch.randn(M, N, device="cuda").bfloat16() if epi in (5, 6, 11) else None)
out2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if epi in (4, 10) else None
layout = os.environ.get("LAYOUT", "sep")
if out2 is not None and layout == "interleave":            # one [M, 2N] buffer, out = left half, out2 = right half of a row
    both = torch.empty(M, 2 * N, device="cuda", dtype=torch.bfloat16)
    out, out2 = both[:, :N], both[:, N:]
elif out2 is not None and layout.startswith("pad"):        # second buffer displaced by `pad<bytes>`
    pad = int(layout[3:])
    raw = torch.empty(2 * M * N + pad // 2 + 64, device="cuda", dtype=torch.bfloat16)
    out = raw[:M * N].view(M, N)
    out2 = raw[M * N + pad // 2:2 * M * N + pad // 2].view(M, N)
if epi == 11:
    bias = None
ops.gemm_nt(a, w, epi, out, bias=bias, out2=out2, aux=aux)
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(reps):
    ops.gemm_nt(a, w, epi, out, bias=bias, out2=out2, aux=aux)
t1.record()
torch.cuda.synchronize()
us = t0.elapsed_time(t1) / reps * 1e3
print(f"{os.environ.get('SAIS_HIP_LIB', 'default')} layout={layout} d={0 if out2 is None else out2.data_ptr() - out.data_ptr()} N={N} K={K} epi={epi}: {us:.1f} us  {2.0 * M * N * K / us / 1e6:.0f} TFLOP/s")
