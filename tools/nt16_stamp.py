#!/usr/bin/env python3
"""Barrier timeline of the anti-phase 16-wave NT GEMM (gemm_nt_w16_kernel, SAIS_NT_W16=1) from a -DSAIS_NT_STAMP build:
    tools/build_variant_file.sh ntstamp gemm -DSAIS_NT_STAMP
    SAIS_NT_W16=1 SAIS_HIP_LIB=tools/bin/ntstamp/libsais_hip.so python tools/nt16_stamp.py 1536 384 10
Workgroup 0, barriers 24..41 (steady state, three half-periods of six intervals), lane 0 of every wave: arrival at and release from
every barrier.  Per interval and group: `work` = release of the previous barrier -> arrival at this one (the interval's own
instructions), `wait` = arrival -> release (waiting for the slowest wave of the workgroup).  Group 0 = waves 0-7, group 1 = 8-15;
with nk = 6 group 0 runs K-steps in intervals 24-29 and 36-41 and epilogue slices in 30-35, group 1 the opposite."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sais_amd import _lib as L, ops  # noqa: E402

M = 50432
N, K, epi = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
bias = None if epi == 11 else torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
aux = torch.randn(M, N, device="cuda").bfloat16() if epi in (5, 6, 11) else None
out2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if epi in (4, 10) else None
for _ in range(3):
    ops.gemm_nt(a, w, epi, out, bias=bias, out2=out2, aux=aux)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (16 * 36))()
lib = L.load()
lib.sais_debug_nt16_stamps.argtypes = [ctypes.c_void_p]
assert lib.sais_debug_nt16_stamps(buf) == 0
st = [[buf[wv * 36 + i] for i in range(36)] for wv in range(16)]
nk = K // 64
print(f"N {N} K {K} epilogue {epi}: shader-clock cycles, workgroup 0, barriers 24..41")
print("interval   role(g0/g1)   g0 work (min-max)  g0 wait      g1 work (min-max)  g1 wait      interval length")
for k in range(1, 18):
    row = []
    for grp in (0, 1):
        ws = range(8 * grp, 8 * grp + 8)
        work = [st[w_][2 * k] - st[w_][2 * k - 1] for w_ in ws]
        wait = [st[w_][2 * k + 1] - st[w_][2 * k] for w_ in ws]
        row.append((min(work), max(work), min(wait), max(wait)))
    length = max(st[w_][2 * k + 1] for w_ in range(16)) - max(st[w_][2 * k - 1] for w_ in range(16))
    b = 24 + k
    phase0 = "K" if (b // nk) % 2 == 0 else "E"
    phase1 = "E" if phase0 == "K" else "K"
    print(f"{b:5d}      {phase0}{b % nk} / {phase1}{b % nk}      {row[0][0]:5d}-{row[0][1]:5d}   {row[0][2]:5d}-{row[0][3]:5d}    "
          f"{row[1][0]:5d}-{row[1][1]:5d}   {row[1][2]:5d}-{row[1][3]:5d}    {length:6d}")
