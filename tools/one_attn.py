#!/usr/bin/env python3
"""Run the ViT attention kernels at config-2 size a few times (for rocprofv3 passes)."""
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
for _ in range(3):
    ops.vit_attn_fwd(qkv, F, out, lse)
    ops.vit_attn_bwd(qkv, dout, out, lse, delta, F, dqkv)
torch.cuda.synchronize()
