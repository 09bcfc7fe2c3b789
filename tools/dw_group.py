"""dW of ten ViT blocks (M = 50 432) as 10 / 5 / 2 / 1 grouped launches of 4 / 8 / 20 / 40 GEMMs (LABNOTES R6.8): time per block and
exactness (small-integer operands).  Every block has its own operand and gradient buffers, as in the step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sais_amd import ops

M = 197 * 256
dev = "cuda"
shapes = [(384, 1536), (1536, 384), (384, 384), (1152, 384)]
g = torch.Generator().manual_seed(0)
blocks = []
for b in range(10):
    items = []
    for n1, n2 in shapes:
        p = torch.randint(-2, 3, (M, n1), generator=g).to(torch.bfloat16).to(dev)
        q = torch.randint(-3, 4, (M, n2), generator=g).to(torch.bfloat16).to(dev)
        items.append((p, q, torch.zeros(n1, n2, device=dev), torch.zeros(n1, device=dev)))
    blocks.append(items)
ref = [[(p.float().t() @ q.float(), p.float().sum(0)) for p, q, _, _ in items] for items in blocks[:2]]


def run(G):
    for i in range(0, 10, G):
        ops.gemm_tn_grouped([it for blk in blocks[i:i + G] for it in blk], M)


for G in (1, 2, 5, 10):
    for blk in blocks:
        for _, _, dW, db in blk:
            dW.zero_(); db.zero_()
    run(G)
    torch.cuda.synchronize()
    ok = all(torch.equal(dW, rw) and torch.equal(db, rb) for blk, rr in zip(blocks[:2], ref) for (_, _, dW, db), (rw, rb) in zip(blk, rr))
    for _ in range(3):
        run(G)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run(G)
    e1.record()
    torch.cuda.synchronize()
    print(f"G={G:2d}: {e0.elapsed_time(e1) / 100 * 1e3:7.1f} us per block  exact={ok}")
