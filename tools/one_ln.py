#!/usr/bin/env python3
"""Times the LN-fused row GEMMs (M = 50 432) with their epilogue READ operands from HBM vs from one L2-resident row
(stride-0 views): how much of the kernel is the epilogue's reads competing with its writes?  usage: one_ln.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sais_amd import ops  # noqa: E402

M, D = 50432, 384
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)


def timeit(fn):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for K in (384, 1536):
    A = rnd(M, K).bfloat16()
    W = (rnd(D, K) * 0.05).bfloat16()
    bias, gamma, beta = rnd(D), 1 + 0.1 * rnd(D), 0.05 * rnd(D)
    resid = rnd(M, D)
    one = rnd(1, D)
    x_out, xn = torch.empty(M, D, device="cuda"), torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    for name, r in (("hbm", resid), ("row0", one.expand(M, D))):
        us = timeit(lambda: ops.gemm_ln_fwd(A, W, bias, r, x_out, xn, gamma, beta, 1e-6, mean, rstd))
        print(f"ln_fwd K={K} resid={name}: {us:.1f} us")
    ops.gemm_ln_fwd(A, W, bias, resid, x_out, xn, gamma, beta, 1e-6, mean, rstd)
    dx32, dx16 = torch.empty(M, D, device="cuda"), torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
    dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    dres = rnd(M, D)
    for name, xx, dr in (("hbm", x_out, dres), ("row0", x_out[:1].expand(M, D), dres[:1].expand(M, D))):
        us = timeit(lambda: ops.gemm_ln_bwd(A, W, xx, mean, rstd, gamma, dres=dr, dx32=dx32, dx16=dx16, dgamma=dg, dbeta=db))
        print(f"ln_bwd K={K} x,dres={name}: {us:.1f} us")
