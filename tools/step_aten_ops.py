#!/usr/bin/env python3
"""Which torch (aten) ops one eager training step issues besides this library's kernels, and from where (TorchDispatchMode +
traceback).  The hand-written kernels are the step; these are the glue launches worth removing."""
import os, sys, traceback, collections, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests", "golden"))
import bench, synth
from sais_amd.parallel import GradSync
from torch.utils._python_dispatch import TorchDispatchMode
dev = torch.device("cuda:0"); B, T, C = 8, 32, 2
vit, model, protos, opt = bench.build(dev, B, T, C, lr=0.1)
frames = synth.clips(seed=0, B=B, T=T).view(B * T, 3, 224, 224).to(dev)
pad = synth.padding_mask([T] * B).to(dev); labels = synth.labels(seed=0, B=B, nclasses=C)
step = bench.make_step(vit, model, protos, opt, GradSync(1, active=False), frames, pad, labels, B, T, 1)
for _ in range(3): step()
torch.cuda.synchronize()
seen = collections.Counter()
VIEW = ("view", "reshape", "_unsafe_view", "as_strided", "slice", "select", "expand", "detach", "alias", "t", "transpose", "permute", "unsqueeze", "squeeze", "_reshape_alias", "split", "unbind", "empty", "empty_like", "empty_strided", "new_empty", "stride", "size", "is_same_size", "lift_fresh", "_local_scalar_dense", "narrow", "unflatten")
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name not in VIEW:
            fr = [f for f in traceback.extract_stack() if "/root/repo/" in f.filename and "tools/" not in f.filename]
            where = f"{fr[-1].filename.split('/root/repo/')[-1]}:{fr[-1].lineno}" if fr else "autograd engine"
            seen[(name, where)] += 1
        return func(*args, **(kwargs or {}))
with Log():
    step()
torch.cuda.synchronize()
for (n, w), c in sorted(seen.items(), key=lambda kv: kv[0][1]):
    print(f"{n:24s} x{c:2d}  {w}")
print("non-view aten ops per step:", sum(seen.values()))
