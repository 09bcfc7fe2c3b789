#!/usr/bin/env python3
"""Throughput of the frame-preprocessing kernel (sais_preprocess_run) on resident uint8 frames, against its HBM
roofline: algorithmic bytes per frame = the crop region read once (0.8 H x 0.8 W x 3) + the fp32 [3,224,224] output."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sais_amd.preprocess import FramePreprocessor  # noqa: E402


def main():
    for (h, w, n) in ((1080, 1920, 128), (720, 1280, 256), (480, 854, 512)):
        frames = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda")
        pre = FramePreprocessor(h, w)
        out = torch.empty(n, 3, 224, 224, device="cuda")
        for _ in range(3):
            pre(frames, out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        iters = 10
        for _ in range(iters):
            pre(frames, out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        l, t, r, b = pre.box
        nbytes = n * ((r - l) * (b - t) * 3 + 3 * 224 * 224 * 4)
        print(f"{h}x{w}: {n} frames in {dt * 1e3:.3f} ms = {n / dt:,.0f} frames/s, {nbytes / dt / 1e12:.2f} TB/s algorithmic "
              f"({nbytes / dt / 8e12 * 100:.0f}% of 8 TB/s)")


if __name__ == "__main__":
    main()
