"""Per-launch time of the temporal attention kernels at the benchmark shape (B = 8 sequences of 33 tokens)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sais_amd import ops
B, S = 8, 33
qkv = torch.randn(B * S, 1152, device="cuda"); pad = torch.zeros(B, S, dtype=torch.uint8, device="cuda")
ctx = torch.empty(B * S, 384, device="cuda"); avg = torch.empty(B, S, S, device="cuda")
dctx = torch.randn(3, B * S, 384, device="cuda"); dqkv = torch.empty(B * S, 1152, device="cuda")
st = ops.rng_state(1, "cuda")
def t(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print("fwd            %.1f us" % t(lambda: ops.temporal_attn_fwd(qkv, pad, B, S, ctx, None)))
print("fwd + map      %.1f us" % t(lambda: ops.temporal_attn_fwd(qkv, pad, B, S, ctx, avg)))
print("fwd + dropout  %.1f us" % t(lambda: ops.temporal_attn_fwd(qkv, pad, B, S, ctx, None, p_drop=0.1, rng=st, site=0)))
print("bwd            %.1f us" % t(lambda: ops.temporal_attn_bwd(qkv, pad, B, S, dctx, dqkv)))
print("bwd + dropout  %.1f us" % t(lambda: ops.temporal_attn_bwd(qkv, pad, B, S, dctx, dqkv, p_drop=0.1, rng=st, site=0)))
# the temporal GEMM shapes of a layer, back to back (compare with the in-step durations of profiles/*kernel_stats.csv)
from sais_amd import _lib as L
M = 264
a384, a2048 = torch.randn(M, 384, device="cuda"), torch.randn(M, 2048, device="cuda")
for name, a, N, epi, ns in (("in_proj  N1152 K384  bias", a384, 1152, L.TG_BIAS, 1), ("linear1  N2048 K384  relu+drop", a384, 2048, L.TG_BIAS_RELU, 1),
                            ("out_proj N384  K384  raw x3", a384, 384, L.TG_RAW, 3), ("linear2  N384  K2048 raw x8", a2048, 384, L.TG_RAW, 8)):
    w = torch.randn(N, a.shape[1], device="cuda") * 0.05
    bias = torch.randn(N, device="cuda")
    out = torch.empty((ns, M, N) if epi == L.TG_RAW else (M, N), device="cuda")
    drop = (0.1, st, 2) if epi == L.TG_BIAS_RELU else None
    print("tgemm %-30s %.1f us" % (name, t(lambda: ops.tgemm(a, w, epi, out, bias=None if epi == L.TG_RAW else bias, nsplit=ns, drop=drop))))
emb = torch.empty(8, 256, device="cuda"); rep = torch.empty(8, 384, device="cuda"); z = torch.randn(8 * 33, 384, device="cuda")
W, bb = torch.randn(256, 384, device="cuda"), torch.randn(256, device="cuda")
print("head_fwd %.1f us" % t(lambda: ops.head_fwd(z, None, 33 * 384, 8, W, bb, rep, emb)))
