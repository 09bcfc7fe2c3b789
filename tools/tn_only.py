#!/usr/bin/env python3
"""Runs only the grouped weight-gradient GEMM of one ViT block (M = 50 432) a few times: a target for rocprofv3."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sais_amd import ops  # noqa: E402

M, D, HID = 50432, 384, 1536
g = torch.Generator(device="cuda").manual_seed(0)
r16 = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
items = []
for n1, n2 in ((D, HID), (HID, D), (D, D), (3 * D, D)):
    items.append((r16(M, n1), r16(M, n2), torch.zeros(n1, n2, device="cuda"), torch.zeros(n1, device="cuda")))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ops.gemm_tn_grouped(items, M)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(n):
    ops.gemm_tn_grouped(items, M)
b.record()
torch.cuda.synchronize()
fl = sum(2.0 * M * p.shape[1] * q.shape[1] for p, q, _, _ in items)
us = a.elapsed_time(b) / n * 1e3
print(f"{os.environ.get('SAIS_HIP_LIB', 'default')}: {us:.1f} us / launch, {fl / us / 1e6:.0f} TFLOP/s")
