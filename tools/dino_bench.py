#!/usr/bin/env python3
"""Times the DINO pre-training step (sais_amd/dino.py) on one MI355X at the reference's default configuration
(main_dino.py get_args_parser: vit_small/16, out_dim 65536, batch_size_per_gpu 64, 2 global 224 x 224 + 8 local 96 x 96
crops, drop_path_rate 0.1, clip_grad 3.0, AdamW) on synthetic crops, and prints ONE JSON line.
    python tools/dino_bench.py [--batch 64] [--local-crops 8] [--out-dim 65536] [--steps 10] [--warmup 3] [--kernels]
--kernels adds a per-kernel table from live HIP-event brackets (sais_amd.ops.KernelTimer) for the MFMA kernels."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sais_amd import dino, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--local-crops", type=int, default=8)
    ap.add_argument("--out-dim", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--drop-path", type=float, default=0.1)
    ap.add_argument("--kernels", action="store_true")
    a = ap.parse_args()
    dev = "cuda:0"
    torch.cuda.set_device(0)
    student, teacher = dino.build_student_teacher(out_dim=a.out_dim, drop_path_rate=a.drop_path, device=dev)
    loss_mod = dino.DINOLoss(a.out_dim, a.local_crops + 2, 0.04, 0.04, 0, 100).to(dev)
    opt = dino.DINOOptimizer(student, teacher)
    n = a.warmup + a.steps + 1
    lr_s = dino.cosine_scheduler(0.0005 * a.batch / 256.0, 1e-6, 100, n, warmup_epochs=0)
    wd_s = dino.cosine_scheduler(0.04, 0.4, 100, n)
    mom_s = dino.cosine_scheduler(0.996, 1, 100, n)
    g = torch.Generator(device=dev).manual_seed(0)
    images = [torch.randn(a.batch, 3, 224, 224, device=dev, generator=g) for _ in range(2)] + \
             [torch.randn(a.batch, 3, 96, 96, device=dev, generator=g) for _ in range(a.local_crops)]

    def step(it):
        return dino.train_step(student, teacher, loss_mod, opt, images, it, 1, lr_s, wd_s, mom_s, clip_grad=3.0,
                               freeze_last_layer=1)[0]

    for it in range(a.warmup):
        loss = step(it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(a.steps):
        loss = step(a.warmup + it)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    out = {"metric": "dino_pretrain_images_per_s", "value": round(a.batch / dt, 1), "unit": "images/s", "n_gpus": 1,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True,
           "dtype": "bf16 MFMA operands / f32 accumulate, master weights and head f32", "data": "synthetic",
           "loss": round(loss.item(), 4),
           "config": {"workload": f"DINO ViT-S/16 step, B={a.batch}, 2x224 + {a.local_crops}x96 crops, out_dim {a.out_dim}",
                      "tokens_per_step": a.batch * (2 * 197 * 2 + a.local_crops * 37 + 0)},
           "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}
    if a.kernels:
        ops.TIMER = ops.KernelTimer()
        step(a.warmup + a.steps)
        torch.cuda.synchronize()
        summ = ops.TIMER.summary()
        ops.TIMER = None
        out["kernels_ms"] = {k: round(v["total_ms"], 3) for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
