#!/usr/bin/env python3
"""In-kernel shader clock of the step's big kernels while the training step replays (MI355X_MICROARCH.md, DVFS give-back item 6).
Needs a -DSAIS_CLK_STAMP build:  tools/build_variant.sh clk -DSAIS_CLK_STAMP ;  SAIS_HIP_LIB=tools/bin/clk/libsais_hip.so python tools/clk_probe.py
Prints, per stamped kernel, workgroup 0's lifetime in us and d(memtime)/d(memrealtime) x 100 MHz."""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench  # noqa: E402
import synth  # noqa: E402
from sais_amd import _lib as L  # noqa: E402
from sais_amd.graph import GraphedStep  # noqa: E402
from sais_amd.parallel import GradSync  # noqa: E402

NAMES = {("gemm", 0): "w8p qkv / plain", ("gemm", 1): "w8p fc1 + GELU / GELU'", ("gemm", 2): "w8p dX fc2 x GELU'", ("gemm", 3): "dW (tn_pp)",
         ("row", 4): "row LN_FWD K384 (proj)", ("row", 5): "row LN_FWD K1536 (fc2)", ("row", 6): "row LN_BWD K1152 (dX qkv)",
         ("row", 7): "row LN_BWD K1536 (dX fc1)", ("row", 8): "row plain K384 (dX proj)", ("row", 9): "row plain K1536",
         ("attn", 10): "attention fwd", ("attn", 11): "attention bwd"}
dev = torch.device("cuda:0")
B, T, C = 8, 32, 2
zero = os.environ.get("ZERO_INPUT") == "1"
vit, model, protos, opt = bench.build(dev, B, T, C, lr=0.1)
frames = synth.clips(seed=0, B=B, T=T).view(B * T, 3, 224, 224).to(dev)
if zero:
    frames.zero_()
pad = synth.padding_mask([T] * B).to(dev)
labels = synth.labels(seed=0, B=B, nclasses=C)
sync = GradSync(1, active=False)
step = bench.make_step(vit, model, protos, opt, sync, frames, pad, labels, B, T, 1)
vit(frames[:2]); model._engine(dev)
g = GraphedStep(step, warmup=2)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < secs:
    for _ in range(10):
        g()
    torch.cuda.synchronize(); n += 10
print(f"{n} steps in {time.perf_counter() - t0:.2f} s = {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step (zero input: {zero})")
lib = L.load()
for f in ("gemm", "row", "attn"):
    buf = (ctypes.c_ulonglong * 32)()
    fn = getattr(lib, "sais_debug_clk_" + f)
    fn.argtypes = [ctypes.c_void_p]
    assert fn(buf) == 0
    for i in range(16):
        dm, dr = buf[2 * i], buf[2 * i + 1]
        if dr:
            print(f"{NAMES.get((f, i), f + str(i)):34s} wg0 lifetime {dr / 100:8.1f} us   clock {dm / dr * 100 / 1e3:6.3f} GHz")
