#!/usr/bin/env python3
"""Where does the 192x384 dW kernel differ from fp32 torch?  Prints the 32x32 blocks of each item whose error is over the bar."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sais_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 12608
g = torch.Generator().manual_seed(5)
shapes = [(384, 1536), (1536, 384), (384, 384), (1152, 384)]
items, refs = [], []
for n1, n2 in shapes:
    p = torch.randn(M, n1, generator=g).to(torch.bfloat16).cuda(); q = torch.randn(M, n2, generator=g).to(torch.bfloat16).cuda()
    items.append((p, q, torch.zeros(n1, n2, device="cuda"), torch.zeros(n1, device="cuda")))
    refs.append((p.float().t() @ q.float(), p.float().sum(0)))
for rep in range(3):
    for it in items: it[2].zero_(); it[3].zero_()
    ops.gemm_tn_grouped(items, M)
    torch.cuda.synchronize()
    for k, ((p, q, dW, db), (rw, rb)) in enumerate(zip(items, refs)):
        err = (dW - rw).abs()
        bad = err > 2e-3 * math.sqrt(M) + 1e-4 * rw.abs()
        eb = (db - rb).abs().max().item()
        n1, n2 = dW.shape
        blk = bad.view(n1 // 32, 32, n2 // 32, 32).sum((1, 3))
        nz = blk.nonzero().tolist()
        print(f"rep {rep} item {k} {n1}x{n2}: bad {int(bad.sum())}, max err {err.max().item():.3g}, db err {eb:.3g}, bad 32x32 blocks (row,col,count): {[(a, b, int(blk[a, b])) for a, b in nz][:24]}")
