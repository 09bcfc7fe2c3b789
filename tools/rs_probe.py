import os, sys, torch
sys.path.insert(0, os.getcwd())
from sais_amd import _lib as L, ops
def t(M, N, K=384, epi=L.EPI_BIAS_BF16):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    f = lambda: ops.gemm_nt(a, w, epi, out, bias=b)
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    ts.sort(); return ts[3] * 1e3
for N in (128, 256, 512, 1152, 2304, 4608):
    print(f"M=32768 N={N:5d}: {t(32768, N):7.1f} us   ({N // 128} tiles per unit)")
