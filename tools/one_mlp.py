#!/usr/bin/env python3
"""Times the MLP branch of a ViT block at the benchmark's size (M = 50 432): the fused kernels (sais_mlp_fwd / sais_mlp_bwd)
against the launch pairs they replace, interleaved in one process (rounds x variants, median and min).
usage: one_mlp.py [rounds] [M]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sais_amd import _lib as L  # noqa: E402
from sais_amd import ops  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
M = int(sys.argv[2]) if len(sys.argv) > 2 else 50432
D, H = 384, 1536
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
e16 = lambda *s: torch.empty(*s, device="cuda", dtype=torch.bfloat16)
e32 = lambda *s: torch.empty(*s, device="cuda")

xn2 = rnd(M, D).bfloat16()
w1, w2 = (rnd(H, D) * 0.06).bfloat16(), (rnd(D, H) * 0.04).bfloat16()
w2t, w1t = w2.t().contiguous(), w1.t().contiguous()
b1, b2 = rnd(H) * 0.3, rnd(D) * 0.1
gamma, beta = 1 + 0.1 * rnd(D), 0.05 * rnd(D)
resid = rnd(M, D) * 2
h, gd = e16(M, H), e16(M, H)
x_out, xn = e32(M, D), e16(M, D)
mean, rstd = e32(M), e32(M)
d16 = (rnd(M, D) * 0.5).bfloat16()
du = e16(M, H)
dx32, dx16 = rnd(M, D), e16(M, D)
dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
# a 600-MB scribble between timed launches: every variant starts from cold caches, as inside the training step
junk = torch.empty(300 << 20, device="cuda", dtype=torch.bfloat16)


def fwd_two():
    ops.gemm_nt(xn2, w1, L.EPI_BIAS_GELU_GRAD_BF16, h, bias=b1, out2=gd)
    ops.gemm_ln_fwd(h, w2, b2, resid, x_out, xn, gamma, beta, 1e-6, mean, rstd)


def fwd_fused():
    ops.mlp_fwd(xn2, w1, b1, w2, b2, resid, x_out, h=h, g=gd, xn_out=xn, gamma=gamma, beta=beta, eps=1e-6, mean=mean, rstd=rstd)


def fwd_fused_inference():
    ops.mlp_fwd(xn2, w1, b1, w2, b2, resid, x_out, xn_out=xn, gamma=gamma, beta=beta, eps=1e-6)


def bwd_two():
    ops.gemm_nt(d16, w2t, L.EPI_MUL_BF16, du, aux=gd)
    ops.gemm_ln_bwd(du, w1t, x_out, mean, rstd, gamma, dres=dx32, dx32=dx32, dx16=dx16, dgamma=dg, dbeta=db)


def bwd_fused():
    ops.mlp_bwd(d16, w2t, gd, w1t, du, x_out, mean, rstd, gamma, dres=dx32, dx32=dx32, dx16=dx16, dgamma=dg, dbeta=db)


variants = dict(fwd_two_launches=fwd_two, fwd_fused=fwd_fused, fwd_fused_inference=fwd_fused_inference,
                bwd_two_launches=bwd_two, bwd_fused=bwd_fused)
for fn in variants.values():
    fn()
torch.cuda.synchronize()
times = {k: [] for k in variants}
for _ in range(rounds):
    for name, fn in variants.items():
        junk.fill_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        times[name].append(a.elapsed_time(b) * 1e3)
flops = 4.0 * M * D * H
for name, t in times.items():
    med, mn = statistics.median(t), min(t)
    print(f"{name:22s} median {med:7.1f} us  min {mn:7.1f} us   {flops / med / 1e6:7.1f} TFLOP/s (median)")
