#!/usr/bin/env python3
"""Phase timeline of the ViT attention backward from the shader-clock stamps of a -DSAIS_ATTN_STAMP build
(tools/build_variant.sh stamp -DSAIS_ATTN_STAMP; SAIS_HIP_LIB=tools/bin/stamp/libsais_hip.so python tools/attn_stamp.py).
Workgroup 0, second problem, lane 0 of every wave.  Stamp indices: 0 problem start, 1 staging done, 2 after the barrier; per query
step qs: 3+5qs P / dS done, 4+5qs dV / dK issued, 5+5qs after the step barrier, 6+5qs dQ stored; 38 loop done, 39 problem done."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sais_amd import _lib as L, ops  # noqa: E402

F = 256
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(F * 197, 1152, device="cuda", generator=g).bfloat16()
dout = torch.randn(F * 197, 384, device="cuda", generator=g).bfloat16()
out = torch.empty(F * 197, 384, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(F, 6, 197, device="cuda")
dqkv = torch.empty(F * 197, 1152, device="cuda", dtype=torch.bfloat16)
ops.vit_attn_fwd(qkv, F, out, lse)
for _ in range(3):
    ops.vit_attn_bwd(qkv, dout, out, lse, None, F, dqkv)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (16 * 40))()
lib = L.load()
lib.sais_debug_attn_stamps.argtypes = [ctypes.c_void_p]
assert lib.sais_debug_attn_stamps(buf) == 0
st = [[buf[w * 40 + i] for i in range(40)] for w in range(16)]
t0 = min(st[w][0] for w in range(16))
print("cycles relative to the problem start (shader clock); waves 0 (key tile), 8 (key tile + dQ), 13 (dQ only)")
for w in (0, 8, 13):
    r = st[w]
    print(f"wave {w:2d}: stage {r[1] - t0:6d}  barrier {r[2] - t0:6d}  loop end {r[38] - t0:6d}  problem end {r[39] - t0:6d}")
    for qs in range(7):
        b = 3 + 5 * qs
        prev = r[2] if qs == 0 else r[5 + 5 * (qs - 1)]
        line = f"   qs {qs}: "
        if w < 13:
            line += f"P/dS +{r[b] - prev:5d}  dV/dK issued +{r[b + 1] - r[b]:5d}  "
            line += f"wait at barrier +{r[b + 2] - r[b + 1]:5d}  "
        else:
            line += f"wait at barrier +{r[b + 2] - prev:5d}  "
        if w >= 8:
            line += f"dQ +{r[b + 3] - r[b + 2]:5d}"
        print(line)
