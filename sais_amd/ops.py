"""Tensor-level wrappers over the C ABI (include/sais_hip.h).  torch is plumbing only: device
memory, the current HIP stream, and dtype checks; every op below is one hand-written gfx950 kernel.
"""
import ctypes
import os
import weakref

import torch

from . import _lib as L

BF16, F32 = torch.bfloat16, torch.float32


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class KernelTimer:
    """Optional live per-kernel timing with HIP events on the launch stream (used by bench.py for the
    roofline object).  tag -> [(start, end, algorithmic flops, algorithmic bytes)]."""

    def __init__(self):
        self.recs = {}

    def run(self, tag, flops, nbytes, fn):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        self.recs.setdefault(tag, []).append((s, e, flops, nbytes))

    def summary(self):
        """Raw HIP-event intervals (an event bracket adds ~4 us to a 30-60 us kernel; rocprofv3's kernel durations in
        profiles/ are the un-bracketed figure).  Tags are "<kernel>[shape]"."""
        out = {}
        for tag, lst in self.recs.items():
            ms = [a.elapsed_time(b) for a, b, _, _ in lst]
            out[tag] = dict(launches=len(lst), total_ms=sum(ms), avg_ms=sum(ms) / len(ms),
                            flops=sum(f for _, _, f, _ in lst) / len(lst), bytes=sum(b for _, _, _, b in lst) / len(lst))
        return out


TIMER = None          # set to a KernelTimer to time the MFMA kernels


def _timed(tag, flops, nbytes, fn):
    if TIMER is None:
        fn()
    else:
        TIMER.run(tag, flops, nbytes, fn)


_NT_NAMES = {12: "raw_slabs_f32", 0: "bias_bf16", 1: "relu_bf16", 2: "f32", 3: "resid_f32", 4: "gelu_bf16", 5: "dgelu_bf16", 6: "drelu_bf16",
             7: "patch_f32", 8: "relu_f32", 9: "drelu_f32", 10: "gelu_grad_bf16", 11: "mul_bf16", 13: "gelu_gradq_bf16",
             14: "mulq_bf16"}


def gelu_grad_q8(M=0):
    """True: the MLP of a ViT block keeps GELU'(u) for its backward as one-byte codes (EPI_BIAS_GELU_GRADQ_BF16 / EPI_MULQ_BF16,
    include/sais_hip.h; half the bytes of the tensor) — the library's default (sais_gelu_grad_bytes() == 1; SAIS_GELU_GRAD_Q8=0
    for bf16).  The one-launch MLP (SAIS_MLP_FUSED=1, an experiment record) has its own bf16 pair."""
    return L.load().sais_gelu_grad_bytes() == 1 and not mlp_fused_enabled(M)


def gelu_grad_buffer(M, n, device):
    return torch.empty(M, n, dtype=torch.uint8 if gelu_grad_q8(M) else BF16, device=device)


def epi_gelu_grad(M=0):
    return L.EPI_BIAS_GELU_GRADQ_BF16 if gelu_grad_q8(M) else L.EPI_BIAS_GELU_GRAD_BF16


def epi_mul(M=0):
    return L.EPI_MULQ_BF16 if gelu_grad_q8(M) else L.EPI_MUL_BF16


def _chk(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda:
        raise L.SaisHipError(f"{name}: expected a device tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise L.SaisHipError(f"{name}: expected {dtype}, got {t.dtype}")
    if t.stride(-1) != 1:
        raise L.SaisHipError(f"{name}: innermost dimension must be contiguous")


def gemm_nt(a, w, epilogue, out, bias=None, out2=None, aux=None, M=None, grp=(0, 0, 0), rowscale=None):
    """out[M,N] = a[M,K] . w[N,K]^T (+epilogue).  a, w bf16 2-D (row stride free).  rowscale f32 [M] (EPI_BIAS_RESID_F32
    only): out = aux + rowscale[m] * (acc + bias), the DropPath form of the residual add."""
    _chk(a, BF16, "A"); _chk(w, BF16, "B"); _chk(bias, F32, "bias"); _chk(rowscale, F32, "rowscale")
    M = a.shape[0] if M is None else M
    N, K = w.shape
    g = L.SaisGemm(_p(a), a.stride(0), _p(w), w.stride(0), M, N, K, epilogue, _p(bias),
                   _p(out), out.stride(-2), _p(out2), 0 if out2 is None else out2.stride(-2),
                   _p(aux), 0 if aux is None else aux.stride(-2), grp[0], grp[1], grp[2], _p(rowscale), 0.0, None, 0)
    nbytes = 2 * (M * K + N * K) + out.element_size() * M * N          # algorithmic: operands once, every output once
    if out2 is not None:
        nbytes += out2.element_size() * M * N
    if aux is not None and epilogue != L.EPI_PATCH_F32:
        nbytes += aux.element_size() * M * N
    _timed(f"gemm_nt<{_NT_NAMES[epilogue]}>[N{N},K{K}]", 2.0 * M * N * K, nbytes,
           lambda: L.call("sais_gemm_nt", ctypes.byref(g), _stream()))
    return out


def gemm_nt_splitk(a, w, ksplit, bias=None, rowscale=None, aux=None, out32=None, out16=None):
    """out = aux + rowscale[m] * (a[M,K] . w[N,K]^T + bias) for SMALL M (the [frames, 384] GEMMs of the CLS-only last block):
    K is cut into `ksplit` slices computed by separate workgroups into raw fp32 slabs, summed in a fixed order by
    sais_splitk_finish (deterministic: no atomics).  out32 f32 and / or out16 bf16."""
    _chk(a, BF16, "A"); _chk(w, BF16, "B"); _chk(bias, F32, "bias"); _chk(rowscale, F32, "rowscale"); _chk(aux, F32, "aux")
    _chk(out32, F32, "out32"); _chk(out16, BF16, "out16")
    M, K = a.shape
    N = w.shape[0]
    ksplit = max(1, min(int(ksplit), K // 64))
    slabs = torch.empty(ksplit, M, N, dtype=F32, device=a.device)
    gemm_nt(a, w, L.EPI_RAW_SLABS_F32, slabs, grp=(ksplit, 0, 0))
    L.call("sais_splitk_finish", _p(slabs), ksplit, M, N, N, _p(bias), _p(rowscale), _p(aux), _ld(aux), _p(out32), _ld(out32),
           _p(out16), _ld(out16), _stream())


ROW_GEMM_MIN_M = 8192        # from this M on, the ViT GEMMs run on the 128 x 384 row-owning kernel (gemm_row.hip)


def _ld(t):
    return 0 if t is None else t.stride(-2)


def gemm_ln_fwd(a, w, bias, resid, x_out, xn_out, gamma, beta, eps, mean=None, rstd=None, rowscale=None):
    """x_out f32[M,384] = a[M,K] . w[384,K]^T + bias + resid ;  xn_out bf16 = LayerNorm(x_out) ; mean/rstd saved.
    One launch (LayerNorm in the GEMM epilogue); resid and x_out may be the same tensor."""
    _chk(a, BF16, "A"); _chk(w, BF16, "W"); _chk(bias, F32, "bias"); _chk(resid, F32, "resid")
    _chk(x_out, F32, "x_out"); _chk(xn_out, BF16, "xn_out"); _chk(gamma, F32, "gamma"); _chk(beta, F32, "beta")
    M, K = a.shape
    if tuple(w.shape) != (384, K):
        raise L.SaisHipError(f"gemm_ln_fwd: W must be [384,{K}], got {tuple(w.shape)}")
    g = L.SaisGemmLn(_p(a), a.stride(0), _p(w), w.stride(0), M, K, _p(bias), _p(resid), _ld(resid), _p(x_out),
                     _ld(x_out), _p(xn_out), _ld(xn_out), _p(gamma), _p(beta), eps, _p(mean), _p(rstd), None, 0,
                     None, None, _p(rowscale), None)
    nbytes = 2 * (M * K + 384 * K) + M * 384 * (4 + 4 + 2)
    _timed(f"gemm_ln_fwd[N384,K{K}]", 2.0 * M * 384 * K, nbytes,
           lambda: L.call("sais_gemm_ln_fwd", ctypes.byref(g), _stream()))


def gemm_ln_bwd(a, w, x, mean, rstd, gamma, dres=None, dx32=None, dx16=None, dgamma=None, dbeta=None, rowscale16=None,
                dres_period=0, xn16=None, beta=None):
    """dy = a[M,K] . w[384,K]^T, then LayerNorm backward at the saved input x / mean / rstd:
    dx = dres + dLN(dy) -> dx32 (f32) and/or dx16 (bf16); dgamma / dbeta accumulated.  dres may alias dx32.
    dres_period > 0: dres is compact [M / period, 384] — row m gets dres[m / period] when m % period == 0, else nothing.
    xn16 (+ beta): the forward's saved bf16 LayerNorm output; the kernel then rebuilds xhat = (xn16 - beta) / gamma from half the
    bytes of x (x stays the fallback for degenerate gamma: include/sais_hip.h, SaisGemmLn.xn16)."""
    _chk(xn16, BF16, "xn16"); _chk(beta, F32, "beta")
    _chk(a, BF16, "A"); _chk(w, BF16, "W"); _chk(x, F32, "x"); _chk(dres, F32, "dres"); _chk(dx32, F32, "dx32")
    _chk(dx16, BF16, "dx16"); _chk(gamma, F32, "gamma")
    M, K = a.shape
    if tuple(w.shape) != (384, K):
        raise L.SaisHipError(f"gemm_ln_bwd: W must be [384,{K}], got {tuple(w.shape)}")
    g = L.SaisGemmLn(_p(a), a.stride(0), _p(w), w.stride(0), M, K, None, _p(x), _ld(x), _p(dx32), _ld(dx32),
                     _p(dx16), _ld(dx16), _p(gamma), None, 0.0, _p(mean), _p(rstd), _p(dres), _ld(dres),
                     _p(dgamma), _p(dbeta), None, _p(rowscale16), int(dres_period),
                     _p(xn16) if (xn16 is not None and beta is not None) else None, 0 if xn16 is None else _ld(xn16))
    g.beta = _p(beta) if beta is not None else None
    nbytes = 2 * (M * K + 384 * K) + M * 384 * ((2 if (xn16 is not None and beta is not None) else 4) + 4 + 4 + 2)
    _timed(f"gemm_ln_bwd[N384,K{K}]", 2.0 * M * 384 * K, nbytes,
           lambda: L.call("sais_gemm_ln_bwd", ctypes.byref(g), _stream()))


MLP_FUSED_MIN_M = 8192       # from this M on, the MLP branch of a ViT block is ONE launch per direction (mlp_fused.hip)


def mlp_fused_enabled(M):
    """SAIS_MLP_FUSED=1 runs the MLP branch of every ViT block as ONE launch per direction (sais_mlp_fwd / _bwd).  It is
    correct (same h / g' bits, same x_out up to fp32 summation order: tests/test_kernels_gpu.py) but MEASURED SLOWER than the
    launch pairs it replaces at the benchmark's size, so the default is the two-launch form."""
    import os
    return os.environ.get("SAIS_MLP_FUSED", "0") == "1" and M >= MLP_FUSED_MIN_M


def mlp_fwd(xn2, w1, b1, w2, b2, resid, x_out, h=None, g=None, xn_out=None, gamma=None, beta=None, eps=1e-6, mean=None,
            rstd=None, rowscale=None):
    """Mlp.forward + residual add + the next LayerNorm in ONE launch (include/sais_hip.h, sais_mlp_fwd):
    u = xn2 . w1^T + b1; h = GELU(u); g = GELU'(u); x_out = resid + rowscale (h . w2^T + b2); xn_out = LN(x_out).
    h / g None: not materialised (inference).  gamma None: no following LayerNorm (the last block)."""
    _chk(xn2, BF16, "xn2"); _chk(w1, BF16, "w1"); _chk(w2, BF16, "w2"); _chk(b1, F32, "b1"); _chk(b2, F32, "b2")
    _chk(resid, F32, "resid"); _chk(x_out, F32, "x_out"); _chk(h, BF16, "h"); _chk(g, BF16, "g"); _chk(xn_out, BF16, "xn_out")
    M, H = xn2.shape[0], w1.shape[0]
    if tuple(w1.shape) != (H, 384) or tuple(w2.shape) != (384, H) or xn2.shape[1] != 384:
        raise L.SaisHipError(f"mlp_fwd: shapes {tuple(xn2.shape)} {tuple(w1.shape)} {tuple(w2.shape)}")
    tail = L.SaisGemmLn(None, 0, None, 0, M, H, _p(b2), _p(resid), _ld(resid), _p(x_out), _ld(x_out), _p(xn_out),
                        _ld(xn_out), _p(gamma), _p(beta), eps, _p(mean), _p(rstd), None, 0, None, None, _p(rowscale), None)
    a = L.SaisMlp(_p(xn2), xn2.stride(0), _p(w1), w1.stride(0), _p(b1), _p(w2), w2.stride(0), M, H, _p(h), _ld(h),
                  _p(g), _ld(g), tail)
    nbytes = 2 * M * 384 + 4 * 384 * H + 2 * M * H * ((h is not None) + (g is not None)) + M * 384 * (4 + 4 + (2 if gamma is not None else 0))
    _timed(f"mlp_fwd[H{H}]", 4.0 * M * 384 * H, nbytes, lambda: L.call("sais_mlp_fwd", ctypes.byref(a), _stream()))


def mlp_bwd(d16, w2t, g, w1t, du, x, mean, rstd, gamma, dres=None, dx32=None, dx16=None, dgamma=None, dbeta=None,
            rowscale16=None):
    """dX of the MLP branch + LayerNorm backward in ONE launch (sais_mlp_bwd): du = (d16 . w2t^T) * g -> du (bf16, kept for
    dW1); dxn = du . w1t^T; dx = dres + dLN(dxn) at the saved x / mean / rstd.  w2t = fc2.weight^T [H,384],
    w1t = fc1.weight^T [384,H]."""
    _chk(d16, BF16, "d16"); _chk(w2t, BF16, "w2t"); _chk(w1t, BF16, "w1t"); _chk(g, BF16, "g"); _chk(du, BF16, "du")
    _chk(x, F32, "x"); _chk(dres, F32, "dres"); _chk(dx32, F32, "dx32"); _chk(dx16, BF16, "dx16"); _chk(gamma, F32, "gamma")
    M, H = d16.shape[0], w2t.shape[0]
    if tuple(w2t.shape) != (H, 384) or tuple(w1t.shape) != (384, H) or d16.shape[1] != 384:
        raise L.SaisHipError(f"mlp_bwd: shapes {tuple(d16.shape)} {tuple(w2t.shape)} {tuple(w1t.shape)}")
    tail = L.SaisGemmLn(None, 0, None, 0, M, H, None, _p(x), _ld(x), _p(dx32), _ld(dx32), _p(dx16), _ld(dx16), _p(gamma),
                        None, 0.0, _p(mean), _p(rstd), _p(dres), _ld(dres), _p(dgamma), _p(dbeta), None, _p(rowscale16))
    a = L.SaisMlp(_p(d16), d16.stride(0), _p(w2t), w2t.stride(0), None, _p(w1t), w1t.stride(0), M, H, _p(du), _ld(du),
                  _p(g), _ld(g), tail)
    nbytes = 2 * M * 384 + 4 * 384 * H + 4 * M * H + M * 384 * (4 + 4 + 4 + 2)
    _timed(f"mlp_bwd[H{H}]", 4.0 * M * 384 * H, nbytes, lambda: L.call("sais_mlp_bwd", ctypes.byref(a), _stream()))


def gemm_nt_f32(a, w, epilogue, out, bias=None, aux=None, M=None, drop=None):
    """fp32-operand variant (bf16x3 split on the matrix cores): out f32[M,N] = a f32[M,K] . w f32[N,K]^T.
    drop = (p, rng_state, site): train-mode dropout fused into the epilogue (include/sais_hip.h, SaisGemm.p_drop)."""
    _chk(a, F32, "A"); _chk(w, F32, "B"); _chk(bias, F32, "bias"); _chk(out, F32, "out"); _chk(aux, F32, "aux")
    M = a.shape[0] if M is None else M
    N, K = w.shape
    # few output tiles: split K so that ~48+ workgroups run (workspace from the caching allocator)
    tiles, nk, ks, ws = (N // 128) * ((M + 127) // 128), K // 64, 1, None
    if tiles < 48:
        for cand in (8, 6, 4, 3, 2):
            if nk % cand == 0 and nk // cand >= 1 and tiles * cand <= 256:
                ks = cand
                break
    if ks > 1:
        ws = torch.empty(ks, M, N, dtype=F32, device=a.device)
    g = L.SaisGemm(_p(a), a.stride(0), _p(w), w.stride(0), M, N, K, epilogue, _p(bias),
                   _p(out), out.stride(-2), _p(ws), ks if ws is not None else 0, _p(aux),
                   0 if aux is None else aux.stride(-2), 0, 0, 0, None,
                   0.0 if drop is None else float(drop[0]), None if drop is None else _p(drop[1]),
                   0 if drop is None else int(drop[2]))
    _timed(f"gemm_nt_f32x3<{_NT_NAMES[epilogue]}>", 2.0 * M * N * K, 4 * (M * K + N * K + M * N),
           lambda: L.call("sais_gemm_nt_f32", ctypes.byref(g), _stream()))
    return out


def tgemm(a, w, epilogue, out, bias=None, aux=None, nsplit=1, drop=None):
    """Temporal-encoder linear layer (include/sais_hip.h, sais_tgemm): out = a f32[M,K] . w f32[N,K]^T on the bf16x3 path.
    epilogue L.TG_RAW: out is f32 [nsplit, M, N] of partial sums (consumed by temporal_ln_fwd / _bwd, temporal_attn_bwd,
    temporal_prepare_bwd); TG_BIAS / TG_BIAS_RELU / TG_DRELU: out f32 [M, N].  drop = (p, rng_state, site)."""
    _chk(a, F32, "A"); _chk(w, F32, "W"); _chk(bias, F32, "bias"); _chk(out, F32, "out"); _chk(aux, F32, "aux")
    M, K = a.shape
    N = w.shape[0]
    g = L.SaisTGemm(_p(a), a.stride(0), _p(w), w.stride(0), M, N, K, epilogue, nsplit, _p(bias), _p(aux),
                    0 if aux is None else aux.stride(0), _p(out), out.stride(-2),
                    0.0 if drop is None else float(drop[0]), None if drop is None else _p(drop[1]),
                    0 if drop is None else int(drop[2]))
    _timed("tgemm", 2.0 * M * N * K, 4 * (M * K + N * K + M * N), lambda: L.call("sais_tgemm", ctypes.byref(g), _stream()))
    return out


def temporal_ln_fwd(slabs, bias, resid, gamma, beta, eps, z, y=None, mean=None, rstd=None, drop=None):
    """y = resid + drop(sum_z slabs[z] + bias) ; z = LayerNorm(y).  slabs f32 [nslab, M, 384]."""
    _chk(slabs, F32, "slabs"); _chk(resid, F32, "resid"); _chk(z, F32, "z"); _chk(y, F32, "y")
    L.call("sais_temporal_ln_fwd", _p(slabs), slabs.shape[0], slabs.stride(0), _p(bias), _p(resid), slabs.shape[1],
           0.0 if drop is None else float(drop[0]), None if drop is None else _p(drop[1]), 0 if drop is None else int(drop[2]),
           _p(y), _p(gamma), _p(beta), eps, _p(z), _p(mean), _p(rstd), _stream())


def temporal_ln_bwd(slabs, add, x, mean, rstd, gamma, dx, dx_drop=None, drop=None, dgamma=None, dbeta=None):
    """dy = sum_z slabs[z] + add ; dx = LayerNorm'(dy) at (x, mean, rstd) ; dx_drop = drop(dx)."""
    _chk(slabs, F32, "slabs"); _chk(add, F32, "add"); _chk(x, F32, "x"); _chk(dx, F32, "dx"); _chk(dx_drop, F32, "dx_drop")
    rows = x.shape[0]
    L.call("sais_temporal_ln_bwd", _p(slabs), 0 if slabs is None else slabs.shape[0], 0 if slabs is None else slabs.stride(0),
           _p(add), _p(x), _p(mean), _p(rstd), _p(gamma), rows, _p(dx), _p(dx_drop),
           0.0 if drop is None else float(drop[0]), None if drop is None else _p(drop[1]), 0 if drop is None else int(drop[2]),
           _p(dgamma), _p(dbeta), _stream())


def gemm_tn(p, q, dW, db=None, nsplit=None):
    """dW[N1,N2] += p[M,N1]^T . q[M,N2] ; db[N1] += colsum(p).  p, q both bf16 or both f32."""
    f32 = p.dtype == F32
    _chk(p, F32 if f32 else BF16, "P"); _chk(q, F32 if f32 else BF16, "Q"); _chk(dW, F32, "dW"); _chk(db, F32, "db")
    M, N1 = p.shape
    N2 = q.shape[1]
    if nsplit is None:
        tiles = (N1 // 128) * (N2 // 128)
        # measured sweep on MI355X (tools/gemm_bench.py): ~430 workgroups is the sweet spot between chip
        # fill (2 workgroups/CU) and fp32-atomic traffic (64 KiB per workgroup)
        nsplit = max(1, min((M + 255) // 256, (432 + tiles - 1) // tiles))
    _timed("gemm_tn_f32" if f32 else "gemm_tn", 2.0 * M * N1 * N2, p.element_size() * M * (N1 + N2) + 4 * N1 * N2,
           lambda: L.call("sais_gemm_tn_f32" if f32 else "sais_gemm_tn", _p(p), p.stride(0), _p(q), q.stride(0), M, N1,
                          N2, _p(dW), dW.stride(0), _p(db), nsplit, _stream()))


def gemm_tn_grouped(items, M, nsplit=None):
    """items: list of (p [M,N1], q [M,N2], dW f32[N1,N2], db f32[N1] | None), p and q all bf16 or all f32; all
    accumulate (+=)."""
    arr = (L.SaisTnItem * len(items))()
    tiles, flops, nbytes = 0, 0.0, 0
    f32 = items[0][0].dtype == F32
    for i, (p, q, dW, db) in enumerate(items):
        _chk(p, F32 if f32 else BF16, "P"); _chk(q, F32 if f32 else BF16, "Q"); _chk(dW, F32, "dW"); _chk(db, F32, "db")
        N1, N2 = p.shape[1], q.shape[1]
        arr[i] = L.SaisTnItem(_p(p), p.stride(0), _p(q), q.stride(0), N1, N2, _p(dW), dW.stride(0), _p(db))
        tiles += (N1 // 128) * (N2 // 128)
        flops += 2.0 * M * N1 * N2
        nbytes += 2 * M * (N1 + N2) + 4 * N1 * N2
    if nsplit is None:
        nsplit = max(1, min((M + 255) // 256, (432 + tiles - 1) // tiles))
    # one tag per distinct launch shape: the full four-GEMM dW of a block, the last block's qkv-only and compact launches and
    # the temporal layers' are different kernels in all but name, and a pooled average would describe none of them
    tag = ("gemm_tn_grouped_f32" if f32 else "gemm_tn_grouped") + ("" if f32 else f"[{len(items)} GEMMs,M{M}]")
    if f32:
        _timed(tag, flops, nbytes, lambda: L.call("sais_gemm_tn_grouped_f32", arr, len(items), M, nsplit, _stream()))
        return
    # large-tile regime (the dW of a ViT block at training size): raw split slabs + a fixed-order finish instead of fp32 atomics
    # (0 bytes: the regime or the slab form does not apply and no workspace is allocated)
    need = L.load().sais_gemm_tn_grouped_slab_bytes(arr, len(items), M)
    ws = _tn_slabs(need, items[0][0].device) if need else None
    _timed(tag, flops, nbytes,
           lambda: L.call("sais_gemm_tn_grouped_ws", arr, len(items), M, nsplit, _p(ws), need, _stream()))


_TN_SLABS = {}


def _tn_slabs(nbytes, device):
    """One slab workspace per device and size class (launches on a stream are ordered, so consecutive dW launches share it);
    allocated outside any hipGraph capture pool by the eager warm-up steps that precede a capture.  A buffer is never freed or
    replaced: a captured graph keeps the pointer it was recorded with, so a larger request gets a NEW buffer beside the old."""
    bufs = _TN_SLABS.setdefault(device, [])
    for buf in bufs:
        if buf.numel() >= nbytes:
            return buf
    buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
    bufs.append(buf)
    bufs.sort(key=lambda b: b.numel())
    return buf


def layernorm_fwd(x, rows, ldx, gamma, beta, eps, y16=None, y32=None, mean=None, rstd=None, ldy16=384, ldy32=384):
    _chk(x, F32, "x"); _chk(y16, BF16, "y16"); _chk(y32, F32, "y32")
    L.call("sais_layernorm_fwd", _p(x), ldx, rows, 384, _p(gamma), _p(beta), eps, _p(y16), ldy16, _p(y32), ldy32,
           _p(mean), _p(rstd), _stream())


def layernorm_bwd(x, ldx, mean, rstd, gamma, rows, dy16=None, dy32=None, dres=None, dx32=None, dx16=None,
                  dgamma=None, dbeta=None, lddy16=384, lddy32=384, lddres=384, lddx32=384, lddx16=384, rowscale16=None,
                  dx32_drop=None, drop=None):
    """dx32_drop (with drop = (p, rng_state, site)): a second fp32 output dropout(dx), same row stride as dx32."""
    _chk(dy16, BF16, "dy16"); _chk(dy32, F32, "dy32"); _chk(dx16, BF16, "dx16"); _chk(dx32, F32, "dx32")
    _chk(dx32_drop, F32, "dx32_drop")
    L.call("sais_layernorm_bwd", _p(dy16), lddy16, _p(dy32), lddy32, _p(x), ldx, _p(mean), _p(rstd), _p(gamma),
           _p(dres), lddres, rows, 384, _p(dx32), lddx32, _p(dx16), lddx16, _p(dgamma), _p(dbeta), _p(rowscale16),
           _p(dx32_drop), 0.0 if drop is None else float(drop[0]), None if drop is None else _p(drop[1]),
           0 if drop is None else int(drop[2]), _stream())


def vit_attn_fwd(qkv, frames, out, lse=None, probs=None, ntok=197):
    _chk(qkv, BF16, "qkv"); _chk(out, BF16, "out")
    _timed("vit_attn_fwd", 4.0 * frames * 6 * ntok * ntok * 64, 2 * frames * ntok * 384 * 4,
           lambda: L.call("sais_vit_attn_fwd", _p(qkv), qkv.stride(0), frames, ntok, _p(out), out.stride(0), _p(lse),
                          _p(probs), _stream()))


def vit_attn_bwd(qkv, dout, out, lse, delta_ws, frames, dqkv, ntok=197):
    """out = the forward attention output (bf16 [frames*ntok, 384]) saved by vit_attn_fwd."""
    _chk(out, BF16, "out")
    _timed("vit_attn_bwd", 10.0 * frames * 6 * ntok * ntok * 64, 2 * frames * ntok * 384 * 9,
           lambda: L.call("sais_vit_attn_bwd", _p(qkv), qkv.stride(0), _p(dout), dout.stride(0), _p(out), out.stride(0),
                          _p(lse), _p(delta_ws), frames, ntok, _p(dqkv), dqkv.stride(0), _stream()))


def vit_block_params(f, i, depth):
    """SaisVitBlockParams of block i from a FlatParams engine (include/sais_hip.h): pointers into the flat parameter /
    gradient / shadow buffers, valid as long as those buffers live."""
    p = f"blocks.{i}."
    nxt = f"blocks.{i + 1}." if i + 1 < depth else None
    P = L.SaisVitBlockParams()
    for k, name in (("qkv", "attn.qkv"), ("proj", "attn.proj"), ("fc1", "mlp.fc1"), ("fc2", "mlp.fc2")):
        setattr(P, k + "_w", _p(f.w(p + name + ".weight")))
        setattr(P, k + "_b", _p(f.w32(p + name + ".bias")))
        setattr(P, k + "_wt", _p(f.wt16[p + name + ".weight"]))
        setattr(P, "d_" + k + "_w", _p(f.g(p + name + ".weight")))
        setattr(P, "d_" + k + "_b", _p(f.g(p + name + ".bias")))
    P.norm1_g, P.norm2_g, P.norm2_b = _p(f.w32(p + "norm1.weight")), _p(f.w32(p + "norm2.weight")), _p(f.w32(p + "norm2.bias"))
    P.norm1_b = _p(f.w32(p + "norm1.bias"))
    P.d_norm1_g, P.d_norm1_b = _p(f.g(p + "norm1.weight")), _p(f.g(p + "norm1.bias"))
    P.d_norm2_g, P.d_norm2_b = _p(f.g(p + "norm2.weight")), _p(f.g(p + "norm2.bias"))
    if nxt is not None:
        P.next_norm_g, P.next_norm_b = _p(f.w32(nxt + "norm1.weight")), _p(f.w32(nxt + "norm1.bias"))
    return P


def block_workspace(op, frames, ntok, device):
    n = L.load().sais_workspace_bytes(op, frames, ntok)
    return torch.empty(n, dtype=torch.uint8, device=device)


def vit_block_fwd(P, frames, ntok, xn1, x_in, qkv, attn_out, lse, x_mid, xn2, mean2, rstd2, h, gelu_grad, x_out, xn_next,
                  mean_next, rstd_next, rs_attn, rs_mlp, ws):
    """One ViT Block forward as ONE C call (sais_vit_block_fwd: C-side sequencing of the GEMM-level launches)."""
    a = L.SaisVitBlockFwd(frames, ntok, _p(xn1), _p(x_in), _p(qkv), _p(attn_out), _p(lse), _p(x_mid), _p(xn2), _p(mean2),
                          _p(rstd2), _p(h), _p(gelu_grad), _p(x_out), _p(xn_next), _p(mean_next), _p(rstd_next), _p(rs_attn),
                          _p(rs_mlp))
    L.call("sais_vit_block_fwd", ctypes.byref(P), ctypes.byref(a), _p(ws), 0 if ws is None else ws.numel(), _stream())


def vit_block_bwd(P, frames, ntok, s, dx, dx16_in, dx16_out, rs_attn, rs_prev, lse, ws, defer_dw=False):
    """One ViT Block backward as ONE C call (sais_vit_block_bwd); s = the tensors the forward saved.  defer_dw: the block's weight /
    bias gradients are left to a later vit_blocks_dw (which takes the returned argument struct); the caller keeps s, ws and dx16_in
    alive and unmodified until then."""
    a = L.SaisVitBlockBwd(frames, ntok, _p(s["x_in"]), _p(s["mean1"]), _p(s["rstd1"]), _p(s["xn1"]), _p(s["qkv"]), _p(s["ao"]),
                          _p(lse), _p(s["x_mid"]), _p(s["mean2"]), _p(s["rstd2"]), _p(s["xn2"]), _p(s["h"]), _p(s["dgelu"]),
                          _p(dx), _p(dx16_in), _p(dx16_out), _p(rs_attn), _p(rs_prev), 1 if defer_dw else 0)
    L.call("sais_vit_block_bwd", ctypes.byref(P), ctypes.byref(a), _p(ws), ws.numel(), _stream())
    return a


def dw_group(dp_hooks=False):
    """ViT blocks per weight-gradient launch at training size (LABNOTES R6.8).  A launch per block cuts M into ten splits to fill
    the chip with its 24 tiles; 24 G tiles need 10 / G, and ten blocks none: 187 -> 160 -> 149 us per block.  Default 10 (one launch
    for blocks 10..1 + the CLS-only block's k / v gradient, one for block 0); 2 when a data-parallel gradient hook waits for the
    blocks' gradients (they become final G blocks at a time: the all-reduce of a bucket overlaps the backward of the next blocks
    only if G is small).  SAIS_DW_GROUP overrides (1 = one launch per block, rounds 1-6); read per call: tests switch it."""
    e = os.environ.get("SAIS_DW_GROUP")
    g = int(e) if e else (2 if dp_hooks else 10)
    return max(1, min(g, (L.TN_MAX_ITEMS - 1) // 4))


def vit_blocks_dw(pending, extra=()):
    """The deferred weight / bias gradients of several blocks as ONE grouped launch (sais_vit_blocks_dw); pending = [(P, a, ws)]
    of vit_block_bwd(..., defer_dw=True) calls; extra = further (p, q, dW, db) items over the same rows."""
    n = len(pending)
    Ps = (ctypes.POINTER(L.SaisVitBlockParams) * n)(*[ctypes.pointer(P) for P, _, _ in pending])
    As = (ctypes.POINTER(L.SaisVitBlockBwd) * n)(*[ctypes.pointer(a) for _, a, _ in pending])
    Ws = (ctypes.c_void_p * n)(*[_p(ws) for _, _, ws in pending])
    ex = (L.SaisTnItem * max(1, len(extra)))()
    for i, (p, q, dW, db) in enumerate(extra):
        ex[i] = L.SaisTnItem(_p(p), p.stride(0), _p(q), q.stride(0), p.shape[1], q.shape[1], _p(dW), dW.stride(0), _p(db))
    L.call("sais_vit_blocks_dw", Ps, As, Ws, min(ws.numel() for _, _, ws in pending), n, ex, len(extra), _stream())


def temporal_layer_params(f, prefix):
    """SaisTemporalLayerParams of the encoder layer whose parameter names start with `prefix` (FlatParams engine)."""
    P = L.SaisTemporalLayerParams()
    for k, name in (("in_proj", "self_attn.in_proj_"), ("out_proj", "self_attn.out_proj."), ("linear1", "linear1."),
                    ("linear2", "linear2.")):
        setattr(P, k + "_w", _p(f.w32(prefix + name + "weight")))
        setattr(P, k + "_b", _p(f.w32(prefix + name + "bias")))
        setattr(P, k + "_wt", _p(f.wt16.get(prefix + name + "weight")))
        setattr(P, "d_" + k + "_w", _p(f.g(prefix + name + "weight")))
        setattr(P, "d_" + k + "_b", _p(f.g(prefix + name + "bias")))
    for k in ("norm1", "norm2"):
        setattr(P, k + "_g", _p(f.w32(prefix + k + ".weight")))
        setattr(P, k + "_b", _p(f.w32(prefix + k + ".bias")))
        setattr(P, "d_" + k + "_g", _p(f.g(prefix + k + ".weight")))
        setattr(P, "d_" + k + "_b", _p(f.g(prefix + k + ".bias")))
    return P


def temporal_layer_fwd(P, B, S, z, pad, qkv, ctx, attn, y1, z1, m1, r1, h, y2, zo, m2, r2, drop, site0, ws):
    """One post-norm TransformerEncoderLayer forward as ONE C call (sais_temporal_layer_fwd).  drop = (p, rng) or None."""
    pd, rng = drop if drop is not None else (0.0, None)
    a = L.SaisTemporalLayerFwd(B, S, _p(z), _p(pad), _p(qkv), _p(ctx), _p(attn), _p(y1), _p(z1), _p(m1), _p(r1), _p(h), _p(y2),
                               _p(zo), _p(m2), _p(r2), float(pd), _p(rng), int(site0))
    L.call("sais_temporal_layer_fwd", ctypes.byref(P), ctypes.byref(a), _p(ws), ws.numel(), _stream())


def temporal_layer_bwd(P, B, S, a, pad, slabs, add, dx_slabs, dx_add, drop, site0, ws, dw_items=None):
    """One encoder layer backward as ONE C call (sais_temporal_layer_bwd); a = the tensors the forward saved; the gradient of
    the layer output comes as raw slabs + add and the gradient of its input leaves the same way (dx_slabs, dx_add).
    dw_items = (SaisTnItem array, first index): the layer's four weight-gradient GEMMs are written there instead of being
    launched (temporal_dw_deferred launches them; `ws` must stay untouched until then)."""
    pd, rng = drop if drop is not None else (0.0, None)
    items_ptr = None
    if dw_items is not None:
        arr, first = dw_items
        items_ptr = ctypes.addressof(arr) + first * ctypes.sizeof(L.SaisTnItem)
    g = L.SaisTemporalLayerBwd(B, S, _p(a["z"]), _p(a["qkv"]), _p(a["ctx"]), _p(a["y1"]), _p(a["m1"]), _p(a["r1"]), _p(a["z1"]),
                               _p(a["h"]), _p(a["y2"]), _p(a["m2"]), _p(a["r2"]), _p(pad), _p(slabs),
                               0 if slabs is None else slabs.shape[0], 0 if slabs is None else slabs.stride(0), _p(add),
                               _p(dx_slabs), _p(dx_add), float(pd), _p(rng), int(site0), items_ptr)
    L.call("sais_temporal_layer_bwd", ctypes.byref(P), ctypes.byref(g), _p(ws), ws.numel(), _stream())


def tn_items(n):
    return (L.SaisTnItem * n)()


def temporal_dw_deferred(arr, n, M):
    """The weight / bias gradients temporal_layer_bwd deferred: ONE launch for up to SAIS_TN_MAX_ITEMS (16) GEMMs."""
    for lo in range(0, n, L.TN_MAX_ITEMS):
        k = min(L.TN_MAX_ITEMS, n - lo)
        first = ctypes.cast(ctypes.addressof(arr) + lo * ctypes.sizeof(L.SaisTnItem), ctypes.POINTER(L.SaisTnItem))
        L.call("sais_gemm_tn_grouped_f32", first, k, M, 1, _stream())


def vit_attn_cls_fwd(qkv, frames, out_c, ntok=197):
    """The last block's attention for the CLS query only: out_c bf16 [frames, 384] (include/sais_hip.h)."""
    _chk(qkv, BF16, "qkv"); _chk(out_c, BF16, "out")
    _timed("vit_attn_cls_fwd", 4.0 * frames * 6 * ntok * 64, 2 * frames * ntok * 768,
           lambda: L.call("sais_vit_attn_cls_fwd", _p(qkv), qkv.stride(0), frames, ntok, _p(out_c), out_c.stride(0), _stream()))


def vit_attn_cls_bwd(qkv, dout_c, frames, dqkv, ntok=197):
    """dqkv bf16 [frames*ntok, 1152] (written entirely) from the compact dout_c bf16 [frames, 384]."""
    _chk(qkv, BF16, "qkv"); _chk(dout_c, BF16, "dout"); _chk(dqkv, BF16, "dqkv")
    _timed("vit_attn_cls_bwd", 10.0 * frames * 6 * ntok * 64, 2 * frames * ntok * (768 + 1152),
           lambda: L.call("sais_vit_attn_cls_bwd", _p(qkv), qkv.stride(0), _p(dout_c), dout_c.stride(0), frames, ntok,
                          _p(dqkv), dqkv.stride(0), _stream()))


def raft_corr_pyramid(f1, f2):
    """f1, f2: f32 [C = 256, H, W] feature maps of ONE frame pair -> (level 0 [H W, ld0] with row stride ld0 >= H W,
    [level 1, 2, 3] as [H W, (H>>l) (W>>l)]).  Level 0 = <f1[:, i], f2[:, j]> / sqrt(C) on the fp32-grade matrix-core GEMM."""
    C, H, W = f1.shape
    HW = H * W
    npad = (HW + 127) // 128 * 128
    a = (f1.reshape(C, HW).t() * (1.0 / C ** 0.5)).contiguous()                  # [HW, C]; 1/16 is exact in fp32
    b = torch.zeros(npad, C, dtype=F32, device=f1.device)
    b[:HW] = f2.reshape(C, HW).t()
    c0 = torch.empty(HW, npad, dtype=F32, device=f1.device)
    gemm_nt_f32(a, b, L.EPI_BIAS_F32, c0)
    lv = [torch.empty(HW, (H >> l) * (W >> l), dtype=F32, device=f1.device) for l in (1, 2, 3)]
    L.call("sais_raft_corr_pool", _p(c0), npad, HW, H, W, _p(lv[0]), _p(lv[1]), _p(lv[2]), _stream())
    return c0, lv


def raft_lookup(pyr, coords, radius=4):
    """pyr: list over the batch of raft_corr_pyramid results (same H, W); coords f32 [B, 2, H, W] -> f32 [B, 4 (2r+1)^2, H, W]."""
    _chk(coords, F32, "coords")
    B, _, H, W = coords.shape
    n = (2 * radius + 1) ** 2
    out = torch.empty(B, 4 * n, H, W, dtype=F32, device=coords.device)
    for b, (c0, lv) in enumerate(pyr):                     # one frame pair per launch: every pair has its own volume
        L.call("sais_raft_lookup", _p(c0), c0.stride(0), _p(lv[0]), _p(lv[1]), _p(lv[2]), _p(coords[b]), 1, H, W, radius,
               _p(out[b]), _stream())
    return out


_CONV_W = {}


def conv2d(x, weight, bias=None, stride=1, padding=0, relu=False):
    """torch.nn.functional.conv2d(x, weight, bias, stride, padding) [-> relu] for f32 NCHW on the matrix cores: per image an
    im2col gather (sais_im2col_f32) and y[Cout, Ho Wo] = [weight | bias] . cols^T on the fp32-grade bf16x3 GEMM
    (sais_gemm_nt_f32).  The padded [Cout, K + 1 -> 64 k] weight matrix is cached per parameter (inference weights are frozen:
    keyed by storage pointer and version)."""
    _chk(x, F32, "x")
    B, C, H, W = x.shape
    Cout, Cin, kh, kw = weight.shape
    if Cin != C:
        raise L.SaisHipError(f"conv2d: {C} input channels, weight expects {Cin}")
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (padding, padding) if isinstance(padding, int) else padding
    Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
    K = C * kh * kw
    ld = (K + 1 + 63) // 64 * 64
    rows = (Ho * Wo + 127) // 128 * 128
    # cache entry = (weak references to the very tensor objects, their versions, the matrix): a freed tensor's address can be
    # handed to another one by the caching allocator, so neither the pointer nor id() alone identifies a weight
    ent = _CONV_W.get(id(weight))
    wm = None
    if ent is not None:
        wref, wver, bref, bver, mat = ent
        if wref() is weight and wver == weight._version and mat.shape[1] == ld and mat.device == x.device and \
                ((bias is None and bref is None) or (bref is not None and bref() is bias and bver == bias._version)):
            wm = mat
    if wm is None:
        wm = torch.zeros(Cout, ld, dtype=F32, device=x.device)
        wm[:, :K] = weight.detach().reshape(Cout, K).float()
        if bias is not None:
            wm[:, K] = bias.detach().float()
        if len(_CONV_W) > 256:
            _CONV_W.clear()
        _CONV_W[id(weight)] = (weakref.ref(weight), weight._version, None if bias is None else weakref.ref(bias),
                               None if bias is None else bias._version, wm)
    x = x.contiguous()
    cols = torch.empty(rows, ld, dtype=F32, device=x.device)
    y = torch.empty(B, Cout, rows, dtype=F32, device=x.device)
    for b in range(B):
        L.call("sais_im2col_f32", _p(x[b]), C, H, W, kh, kw, sh, sw, ph, pw, _p(cols), ld, rows, _stream())
        gemm_nt_f32(wm, cols, L.EPI_BIAS_RELU_F32 if relu else L.EPI_BIAS_F32, y[b])
    return y[:, :, :Ho * Wo].reshape(B, Cout, Ho, Wo)


def touch(t):
    """Prefetch hint: pull a contiguous tensor towards the GPU's caches (include/sais_hip.h, sais_touch)."""
    if t is not None and t.numel() and t.is_contiguous():
        L.call("sais_touch", _p(t), t.numel() * t.element_size(), _stream())


def patchify(frames_f32, patches):
    """frames f32 [F,3,side,side] -> bf16 [F*(side/16)^2, 768]."""
    _chk(frames_f32, F32, "frames")
    L.call("sais_patchify", _p(frames_f32), frames_f32.shape[0], frames_f32.shape[-1], _p(patches), _stream())


def vit_cls_rows(cls, pos0, tokens, frames, ntok=197):
    L.call("sais_vit_cls_rows", _p(cls), _p(pos0), _p(tokens), ntok * 384, frames, 384, _stream())


def vit_embed_bwd(dtokens, frames, dcls, dpos, dpatch, ntok=197):
    L.call("sais_vit_embed_bwd", _p(dtokens), frames, ntok, 384, _p(dcls), _p(dpos), _p(dpatch), _stream())


def sgd_step(param, grad, shadow, lr, grad_scale=1.0):
    L.call("sais_sgd_step", _p(param), _p(grad), _p(shadow), param.numel(), lr, grad_scale, _stream())


def cast_bf16(src, dst):
    L.call("sais_cast_bf16", _p(src), _p(dst), src.numel(), _stream())


def transpose_cast_bf16(src, rows, cols, dst):
    L.call("sais_transpose_cast_bf16", _p(src), rows, cols, _p(dst), _stream())


def transpose_f32(src, rows, cols, dst):
    L.call("sais_transpose_f32", _p(src), rows, cols, _p(dst), _stream())


def transpose_table(entries, device):
    """Device descriptor table for transpose_batch: entries = [(src f32 [rows,cols], dst [cols,rows])].
    Layout = SaisTransposeItem (include/sais_hip.h): two pointers, rows, cols, tile_begin, reserved = 4 x int64."""
    import numpy as np
    tab = np.zeros((len(entries), 4), np.int64)
    tiles = 0
    for i, (src, dst) in enumerate(entries):
        rows, cols = src.shape
        assert tuple(dst.shape) == (cols, rows) and src.dtype == F32 and src.is_contiguous() and dst.is_contiguous()
        tab[i, 0], tab[i, 1] = src.data_ptr(), dst.data_ptr()
        tab[i, 2] = rows | (cols << 32)
        tab[i, 3] = tiles
        tiles += ((rows + 31) // 32) * ((cols + 31) // 32)
    return torch.from_numpy(tab).to(device), len(entries), tiles


def transpose_batch(table, nitems, tiles, dst_is_f32):
    L.call("sais_transpose_batch", _p(table), nitems, tiles, 1 if dst_is_f32 else 0, _stream())


def scale_(t, s):
    L.call("sais_scale_f32", _p(t), t.numel(), s, _stream())


def temporal_prepare_fwd(x, clip_stride, frame_stride, pos, cls, B, T, z32, z16):
    L.call("sais_temporal_prepare_fwd", _p(x), clip_stride, frame_stride, _p(pos), _p(cls), B, T, _p(z32), _p(z16),
           _stream())


def temporal_prepare_bwd(dz32, slabs, B, T, dx, clip_stride, frame_stride, accumulate, dpos, dcls):
    """dz = dz32 (f32 [B*(T+1),384] or None) + sum of the raw split-K slabs (f32 [nslab, B*(T+1), 384] or None)."""
    _chk(dz32, F32, "dz32"); _chk(slabs, F32, "slabs")
    L.call("sais_temporal_prepare_bwd", _p(dz32), _p(slabs), 0 if slabs is None else slabs.shape[0],
           0 if slabs is None else slabs.stride(0), B, T, _p(dx), clip_stride, frame_stride,
           1 if accumulate else 0, _p(dpos), _p(dcls), _stream())


def temporal_attn_fwd(qkv, key_pad_u8, B, S, ctx, attn_avg=None, p_drop=0.0, rng=None, site=0):
    L.call("sais_temporal_attn_fwd", _p(qkv), _p(key_pad_u8), B, S, _p(ctx), _p(attn_avg), float(p_drop), _p(rng), site,
           _stream())


def temporal_attn_bwd(qkv, key_pad_u8, B, S, dctx, dqkv, p_drop=0.0, rng=None, site=0):
    """dctx: f32 [B*S,384], or the raw split-K slabs [nslab, B*S, 384] of the out_proj dX GEMM (summed on load)."""
    nslab, stride = (dctx.shape[0], dctx.stride(0)) if dctx.dim() == 3 else (1, 0)
    L.call("sais_temporal_attn_bwd", _p(qkv), _p(key_pad_u8), B, S, _p(dctx), nslab, stride, _p(dqkv), float(p_drop),
           _p(rng), site, _stream())


# ---- train-mode dropout: rng = int64 device tensor {seed, offset} (include/sais_hip.h)
def rng_state(seed, device):
    return torch.tensor([int(seed), 0], dtype=torch.int64, device=device)


def rng_advance(rng):
    L.call("sais_rng_advance", _p(rng), _stream())


def dropout(x, p, rng, site, resid=None, out=None):
    """out = (resid or 0) + x * keep / (1 - p); in place on x unless `out` is given.  Applying it to a gradient with the
    same (rng, site) is its backward."""
    _chk(x, F32, "x"); _chk(resid, F32, "resid")
    out = x if out is None else out
    L.call("sais_dropout_f32", _p(x), _p(resid), _p(out), x.numel(), float(p), _p(rng), site, _stream())
    return out


def droppath_scales(rates_dev, samples, rows_per_sample, rng, site0=0):
    """f32 [nbranch, samples * rows_per_sample]: per-row keep / (1 - rate) of every DropPath branch (one draw per sample)."""
    nb = rates_dev.numel()
    out = torch.empty(nb, samples * rows_per_sample, dtype=F32, device=rates_dev.device)
    L.call("sais_droppath_scales", _p(out), _p(rates_dev), nb, samples, rows_per_sample, _p(rng), site0, _stream())
    return out


def cast_bf16_rows(src, rowscale, dst):
    _chk(src, F32, "src"); _chk(rowscale, F32, "rowscale"); _chk(dst, BF16, "dst")
    L.call("sais_cast_bf16_rows", _p(src), _p(rowscale), _p(dst), src.shape[0], src.shape[1], _stream())


def dropout_mask(n, p, rng, site, device):
    m = torch.empty(n, dtype=torch.uint8, device=device)
    L.call("sais_dropout_mask", _p(m), n, float(p), _p(rng), site, _stream())
    return m


def head_fwd(z_rgb, z_flow, clip_stride, B, W, bias, rep, emb, clip_stride_flow=None, nsnippets=1, second=None):
    """second = (use_b u8 [B], WB, biasB): clips flagged 1 go through linearB (multi-domain models)."""
    csf = clip_stride if clip_stride_flow is None else clip_stride_flow
    ub, WB, bB = second if second is not None else (None, None, None)
    L.call("sais_head_fwd", _p(z_rgb), _p(z_flow), clip_stride, csf, B, nsnippets, _p(W), _p(bias), _p(ub), _p(WB), _p(bB),
           _p(rep), _p(emb), _stream())


def head_bwd(demb, W, rep, z_rgb, z_flow, clip_stride, B, dW, dbias, dz_rgb, dz_flow, clip_stride_flow=None, nsnippets=1,
             second=None):
    """second = (use_b u8 [B], WB, dWB, dbiasB)."""
    csf = clip_stride if clip_stride_flow is None else clip_stride_flow
    ub, WB, dWB, dbB = second if second is not None else (None, None, None, None)
    L.call("sais_head_bwd", _p(demb), _p(W), _p(rep), _p(z_rgb), _p(z_flow), clip_stride, csf, B, nsnippets, _p(ub), _p(WB),
           _p(dW), _p(dbias), _p(dWB), _p(dbB), _p(dz_rgb), _p(dz_flow), _stream())


def mil_forward(z_rgb, seq_stride, clip_pos, B, nsnippets, tokens):
    L.call("sais_mil_forward", _p(z_rgb), seq_stride, _p(clip_pos), B, nsnippets, _p(tokens), _stream())


def mil_head(enc, B, nsnippets, nclasses, WA, bA, WB, bB, w_att, b_att, w_fin, b_fin, reps, logits, attention):
    L.call("sais_mil_head", _p(enc), B, nsnippets, nclasses, _p(WA), _p(bA), _p(WB), _p(bB), _p(w_att), _p(b_att),
           _p(w_fin), _p(b_fin), _p(reps), _p(logits), _p(attention), _stream())


def importance_fwd(z, w, b, M, out):
    L.call("sais_importance_fwd", _p(z), _p(w), _p(b), M, _p(out), _stream())


def importance_bwd(dlogit, z, w, M, dz, dw, db):
    L.call("sais_importance_bwd", _p(dlogit), _p(z), _p(w), M, _p(dz), _p(dw), _p(db), _stream())


def importance_loss(logits, target, ipad_u8, labels_i32, B, T, loss=None, dlogits=None, scale=1.0):
    L.call("sais_importance_loss", _p(logits), _p(target), _p(ipad_u8), _p(labels_i32), B, T, _p(loss), _p(dlogits),
           scale, _stream())


def nce(emb, protos, label_col, sim=None, probs=None, loss=None, demb=None, dprotos=None, loss_scale=1.0):
    B, C = emb.shape[0], protos.shape[0]
    L.call("sais_nce", _p(emb), _p(protos), _p(label_col), B, C, _p(sim), _p(probs), _p(loss), _p(demb), _p(dprotos),
           loss_scale, _stream())


# --------------------------------------------------------------------------- DINO pre-training objective (dino.hip)
def dino_row_lse(x, scale, center=None, out=None):
    """lse[r] = log sum_k exp((x[r, k] - center[k]) * scale); x f32 [rows, n]."""
    _chk(x, F32, "x"); _chk(center, F32, "center")
    out = torch.empty(x.shape[0], dtype=F32, device=x.device) if out is None else out
    L.call("sais_dino_row_lse", _p(x), x.stride(0), x.shape[0], x.shape[1], float(scale), _p(center), _p(out), _stream())
    return out


def dino_loss(student, teacher, center, s_lse, t_lse, B, ncrops, student_temp, teacher_temp, dlogits, loss, partials=None):
    _chk(student, F32, "student"); _chk(teacher, F32, "teacher"); _chk(center, F32, "center"); _chk(dlogits, F32, "dlogits")
    n = student.shape[1]
    if tuple(student.shape) != (ncrops * B, n) or tuple(teacher.shape) != (2 * B, n) or dlogits.shape != student.shape:
        raise L.SaisHipError(f"dino_loss: student {tuple(student.shape)} / teacher {tuple(teacher.shape)} do not match "
                             f"B={B}, ncrops={ncrops}")
    if partials is None:
        partials = torch.empty(L.load().sais_dino_loss_partials(B, n), dtype=F32, device=student.device)
    L.call("sais_dino_loss", _p(student), student.stride(0), _p(teacher), teacher.stride(0), _p(center), _p(s_lse),
           _p(t_lse), B, ncrops, n, float(student_temp), float(teacher_temp), _p(dlogits), dlogits.stride(0), _p(partials),
           _p(loss), _stream())


def dino_colsum(x, out):
    _chk(x, F32, "x"); _chk(out, F32, "out")
    L.call("sais_dino_colsum", _p(x), x.stride(0), x.shape[0], x.shape[1], _p(out), _stream())


def dino_center_ema(center, colsum, momentum, inv_count):
    _chk(center, F32, "center"); _chk(colsum, F32, "colsum")
    L.call("sais_dino_center_ema", _p(center), _p(colsum), center.numel(), float(momentum), float(inv_count), _stream())


def split_bf16x3(src, dst, b_side):
    """dst bf16 [rows, 3 K] = [hi | hi | lo] (A side) or [hi | lo | hi] (B side) of src f32 [rows, K]."""
    _chk(src, F32, "src"); _chk(dst, BF16, "dst")
    L.call("sais_split_bf16x3", _p(src), src.stride(0), src.shape[0], src.shape[1], _p(dst), 1 if b_side else 0, _stream())


def gelu_fwd_f32(u, h):
    _chk(u, F32, "u"); _chk(h, F32, "h")
    L.call("sais_gelu_fwd_f32", _p(u), _p(h), u.numel(), _stream())


def gelu_bwd_f32(dh, u, du):
    _chk(dh, F32, "dh"); _chk(u, F32, "u"); _chk(du, F32, "du")
    L.call("sais_gelu_bwd_f32", _p(dh), _p(u), _p(du), u.numel(), _stream())


def l2norm_fwd(z, out, inv, eps=1e-12):
    _chk(z, F32, "z"); _chk(out, F32, "out")
    L.call("sais_l2norm_fwd", _p(z), z.shape[0], z.shape[1], eps, _p(out), _p(inv), _stream())


def l2norm_bwd(dout, out, inv, dz, eps=1e-12):
    _chk(dout, F32, "dout"); _chk(out, F32, "out"); _chk(dz, F32, "dz")
    L.call("sais_l2norm_bwd", _p(dout), _p(out), _p(inv), out.shape[0], out.shape[1], eps, _p(dz), _stream())


def weight_norm_fwd(v, g, w, inv):
    _chk(v, F32, "v"); _chk(g, F32, "g"); _chk(w, F32, "w")
    L.call("sais_weight_norm_fwd", _p(v), _p(g), v.shape[0], v.shape[1], _p(w), _p(inv), _stream())


def weight_norm_bwd(dw, v, g, inv, dv, dg=None):
    _chk(dw, F32, "dw"); _chk(v, F32, "v"); _chk(dv, F32, "dv")
    L.call("sais_weight_norm_bwd", _p(dw), _p(v), _p(g), _p(inv), v.shape[0], v.shape[1], _p(dv), _p(dg), _stream())


def pos_interp_fwd(Wm, pos, out):
    """out [1 + nout, dim] from pos [1 + nin, dim] through the fixed map Wm f32 [nout, nin]."""
    _chk(Wm, F32, "Wm"); _chk(pos, F32, "pos"); _chk(out, F32, "out")
    L.call("sais_pos_interp_fwd", _p(Wm), Wm.shape[0], Wm.shape[1], _p(pos), pos.shape[-1], _p(out), _stream())


def pos_interp_bwd(Wm, dout, dpos):
    _chk(Wm, F32, "Wm"); _chk(dout, F32, "dout"); _chk(dpos, F32, "dpos")
    L.call("sais_pos_interp_bwd", _p(Wm), Wm.shape[0], Wm.shape[1], _p(dout), dout.shape[-1], _p(dpos), _stream())


def grad_norms(grad, chunks, nchunks, seg_first, nseg, partial, norms, scale=1.0):
    L.call("sais_grad_norms", _p(grad), _p(chunks), nchunks, _p(seg_first), nseg, float(scale), _p(partial), _p(norms), _stream())


def adamw_ema_step(desc):
    L.call("sais_adamw_ema_step", ctypes.byref(desc), _stream())
