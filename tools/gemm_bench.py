#!/usr/bin/env python3
"""Micro-benchmark of the MFMA GEMM kernels on the ViT shapes of BASELINE config 2 (M = 50 432).
Random (not zero) operands; interleaved rounds in one process; reports median TFLOP/s per shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sais_amd import _lib as L  # noqa: E402
from sais_amd import ops  # noqa: E402

M = 50432
NT = [("qkv      N1152 K384 ", 1152, 384, L.EPI_BIAS_BF16), ("proj     N384  K384 ", 384, 384, L.EPI_BIAS_RESID_F32),
      ("fc1+gelu N1536 K384 ", 1536, 384, L.EPI_BIAS_GELU_GRAD_BF16), ("fc2+res  N384  K1536", 384, 1536, L.EPI_BIAS_RESID_F32),
      ("dX fc2   N1536 K384 ", 1536, 384, L.EPI_MUL_BF16), ("dX fc1   N384  K1536", 384, 1536, L.EPI_BIAS_BF16),
      ("dX qkv   N384  K1152", 384, 1152, L.EPI_BIAS_BF16), ("dX proj  N384  K384 ", 384, 384, L.EPI_BIAS_BF16)]
TN = [("dW qkv  1152x384 ", 1152, 384), ("dW proj 384x384  ", 384, 384), ("dW fc1  1536x384 ", 1536, 384),
      ("dW fc2  384x1536 ", 384, 1536)]


def timeit(fn, rounds=7):
    ts = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    r16 = lambda *s: (torch.randn(*s, device=dev, generator=g)).to(torch.bfloat16)
    tot_f, tot_t = 0.0, 0.0
    for name, N, K, epi in NT:
        a, w, bias = r16(M, K), r16(N, K) * 0.05, torch.randn(N, device=dev)
        f32 = epi in (L.EPI_BIAS_RESID_F32,)
        out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
        aux = torch.randn(M, N, device=dev) if f32 else (r16(M, N) if epi == L.EPI_MUL_BF16 else None)
        out2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == L.EPI_BIAS_GELU_GRAD_BF16 else None
        fn = lambda: ops.gemm_nt(a, w, epi, out, bias=bias, out2=out2, aux=aux)
        fn(); ms = timeit(fn)
        fl = 2.0 * M * N * K
        tot_f += fl; tot_t += ms
        print(f"NT {name}: {ms * 1e3:7.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s")
    for name, N1, N2 in TN:
        p, q = r16(M, N1), r16(M, N2)
        dW, db = torch.zeros(N1, N2, device=dev), torch.zeros(N1, device=dev)
        fn = lambda: ops.gemm_tn(p, q, dW, db)
        fn(); ms = timeit(fn)
        fl = 2.0 * M * N1 * N2
        tot_f += fl; tot_t += ms
        print(f"TN {name}: {ms * 1e3:7.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s")
    print(f"per-block GEMM set: {tot_t:.3f} ms, {tot_f / tot_t / 1e9:.1f} TFLOP/s  (x12 blocks = {12 * tot_t:.2f} ms/step)")


if __name__ == "__main__":
    main()
