#!/usr/bin/env python3
"""Feature-extraction throughput (frozen ViT-S/16 forward, hipGraph replay of fixed 64-frame batches): the inference
half of the SAIS pipeline (extract_representations.py), inputs resident in HBM."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sais_amd.inference import FeatureExtractor  # noqa: E402
from sais_amd.vit import vit_small  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
vit = vit_small().to(dev).eval()
for bs in (64, 256):
    fx = FeatureExtractor(vit, batch_size=bs, use_graph=True)
    frames = torch.randn(2048, 3, 224, 224, device=dev)
    fx(frames[:bs])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = fx(frames)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"batch {bs}: {frames.shape[0] / dt:,.0f} frames/s ({dt * 1e3 / (frames.shape[0] / bs):.2f} ms per {bs}-frame batch)")
