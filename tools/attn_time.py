#!/usr/bin/env python3
"""Median launch time of the three ViT attention kernels at config-2 size (256 frames)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sais_amd import ops  # noqa: E402

F = 256
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(F * 197, 1152, device="cuda", generator=g).bfloat16()
dout = torch.randn(F * 197, 384, device="cuda", generator=g).bfloat16()
out = torch.empty(F * 197, 384, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(F, 6, 197, device="cuda")
delta = torch.empty(F, 6, 197, device="cuda")
dqkv = torch.empty(F * 197, 1152, device="cuda", dtype=torch.bfloat16)


def t(f, n=20):
    f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


print(f"attn fwd {t(lambda: ops.vit_attn_fwd(qkv, F, out, lse)):.1f} us   attn bwd (dq + dkv) "
      f"{t(lambda: ops.vit_attn_bwd(qkv, dout, out, lse, delta, F, dqkv)):.1f} us")
