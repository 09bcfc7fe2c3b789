#!/usr/bin/env python3
"""Yardsticks on the same MI355X, NOT part of the product or of bench.py's contract:

  --gemm   torch.matmul (hipBLASLt / rocBLAS) in bf16 on the GEMM shapes of one ViT-S block at M = 50 432
           -> what AMD's tuned library reaches where sais_gemm_nt / sais_gemm_tn run;
  --step   the reference's software stack restated with stock torch modules (nn.Linear / SDPA / nn.LayerNorm /
           nn.TransformerEncoder-style temporal layers), eager, fwd + bwd + SGD on config 2 (8 clips x 32 frames),
           fp32 and bf16-autocast -> "the reference's way of running this path" on this GPU.

    python tools/torch_baseline.py --gemm --step
"""
import argparse
import json
import time

import torch
import torch.nn as nn
import torch.nn.functional as F


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def gemm_yardstick():
    M, dev = 50432, "cuda"
    out = []
    for name, n, k in (("qkv", 1152, 384), ("proj", 384, 384), ("fc1", 1536, 384), ("fc2", 384, 1536)):
        a = torch.randn(M, k, device=dev, dtype=torch.bfloat16)
        w = torch.randn(n, k, device=dev, dtype=torch.bfloat16)
        b = torch.randn(n, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: F.linear(a, w, b))
        out.append({"gemm": "nt_" + name, "M": M, "N": n, "K": k, "us": t * 1e6, "tflops": 2 * M * n * k / t / 1e12})
        g = torch.randn(M, n, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: g.t() @ a)                                       # dW = dY^T X  [n,k]
        out.append({"gemm": "tn_" + name, "M": M, "N": n, "K": k, "us": t * 1e6, "tflops": 2 * M * n * k / t / 1e12})
        t = timeit(lambda: g @ w)                                           # dX = dY W    [M,k]
        out.append({"gemm": "nn_" + name, "M": M, "N": n, "K": k, "us": t * 1e6, "tflops": 2 * M * n * k / t / 1e12})
    return out


class Block(nn.Module):
    def __init__(self, d=384, h=6):
        super().__init__()
        self.h = h
        self.norm1, self.norm2 = nn.LayerNorm(d, eps=1e-6), nn.LayerNorm(d, eps=1e-6)
        self.qkv, self.proj = nn.Linear(d, 3 * d), nn.Linear(d, d)
        self.fc1, self.fc2 = nn.Linear(d, 4 * d), nn.Linear(4 * d, d)

    def forward(self, x):
        B, N, D = x.shape
        q, k, v = self.qkv(self.norm1(x)).reshape(B, N, 3, self.h, D // self.h).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, D)
        x = x + self.proj(a)
        return x + self.fc2(F.gelu(self.fc1(self.norm2(x))))


class ViT(nn.Module):
    def __init__(self, d=384, depth=12):
        super().__init__()
        self.patch = nn.Conv2d(3, d, 16, 16)
        self.cls = nn.Parameter(torch.zeros(1, 1, d))
        self.pos = nn.Parameter(torch.zeros(1, 197, d))
        self.blocks = nn.ModuleList(Block(d) for _ in range(depth))
        self.norm = nn.LayerNorm(d, eps=1e-6)

    def forward(self, x):
        x = self.patch(x).flatten(2).transpose(1, 2)
        x = torch.cat((self.cls.expand(x.shape[0], -1, -1), x), 1) + self.pos
        for b in self.blocks:
            x = b(x)
        return self.norm(x)[:, 0]


class Temporal(nn.Module):
    def __init__(self, d=384):
        super().__init__()
        layer = nn.TransformerEncoderLayer(d, 4, 2048, 0.0, batch_first=False)
        self.enc = nn.TransformerEncoder(layer, 4, enable_nested_tensor=False)
        self.cls = nn.Parameter(torch.zeros(1, d))
        self.pos = nn.Parameter(torch.zeros(2000, d))
        self.linear = nn.Linear(d, 256)
        self.protos = nn.Parameter(torch.randn(2, 256))

    def encode(self, x):                                                   # x [B,T,D]
        B, T, D = x.shape
        x = torch.cat((self.cls.expand(B, 1, D), x), 1) + self.pos[:T + 1]
        return self.enc(x.transpose(0, 1))[0]

    def forward(self, x, f, y):
        e = F.relu(self.linear(F.relu(self.encode(x) + self.encode(f))))
        s = F.normalize(e, dim=1) @ F.normalize(self.protos, dim=1).T
        return F.cross_entropy(s, y)


def step_yardstick(dtype, steps=10, warm=3):
    dev = "cuda"
    torch.manual_seed(0)
    vit, tmp = ViT().to(dev), Temporal().to(dev)
    params = list(vit.parameters()) + list(tmp.parameters())
    opt = torch.optim.SGD(params, lr=0.1)
    B, T = 8, 32
    frames = torch.randn(B * T, 3, 224, 224, device=dev)
    flows = torch.randn(B, T // 8, 384, device=dev)
    y = torch.randint(0, 2, (B,), device=dev)

    def one():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=dtype, enabled=dtype != torch.float32):
            reps = vit(frames).float().reshape(B, T, 384)
            loss = tmp(reps, flows, y)
        loss.backward()
        opt.step()
    t = timeit(one, steps, warm)
    return {"dtype": str(dtype).replace("torch.", ""), "ms_per_step": t * 1e3, "frames_per_s": B * T / t}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gemm", action="store_true")
    ap.add_argument("--step", action="store_true")
    a = ap.parse_args()
    res = {}
    if a.gemm:
        res["gemm"] = gemm_yardstick()
        for r in res["gemm"]:
            print("%-9s N=%4d K=%4d  %7.1f us  %6.1f TFLOP/s" % (r["gemm"], r["N"], r["K"], r["us"], r["tflops"]))
    if a.step:
        res["step"] = [step_yardstick(torch.float32), step_yardstick(torch.bfloat16)]
        for r in res["step"]:
            print("torch eager %-8s %8.2f ms/step  %8.0f frames/s" % (r["dtype"], r["ms_per_step"], r["frames_per_s"]))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
