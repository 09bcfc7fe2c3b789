#!/usr/bin/env python3
"""Phase timeline of one tile of the persistent eight-wave NT GEMM from the shader-clock stamps of a -DSAIS_NT_STAMP build
(tools/build_variant.sh ntstamp -DSAIS_NT_STAMP; SAIS_HIP_LIB=tools/bin/ntstamp/libsais_hip.so python tools/nt_stamp.py N K epi).
Workgroup 0, third tile, lane 0 of every wave.  Stamps: 0 tile start; 1 + 2 kt: K-step kt's MFMAs issued and the next operands
awaited; 2 + 2 kt: after its barrier; 13 epilogue done (stores issued); 14 next tile's first operands landed; 15 after the barrier."""
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
buf = (ctypes.c_ulonglong * (8 * 16))()
lib = L.load()
lib.sais_debug_nt_stamps.argtypes = [ctypes.c_void_p]
assert lib.sais_debug_nt_stamps(buf) == 0
nk = K // 64
print(f"N {N} K {K} epilogue {epi}: cycles per phase (shader clock), third tile of workgroup 0")
for wv in range(8):
    r = [buf[wv * 16 + i] for i in range(16)]
    ks = "  ".join(f"k{kt}: +{r[1 + 2 * kt] - (r[0] if kt == 0 else r[2 * kt]):4d} wait +{r[2 + 2 * kt] - r[1 + 2 * kt]:4d}" for kt in range(nk))
    print(f"wave {wv}: {ks}   epilogue +{r[13] - r[2 * nk]:5d}  operands +{r[14] - r[13]:5d}  barrier +{r[15] - r[14]:4d}   tile {r[15] - r[0]:6d}")
