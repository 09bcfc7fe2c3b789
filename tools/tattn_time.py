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
