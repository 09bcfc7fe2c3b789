"""Per-kernel parity on a real MI355X: every entry point of libsais_hip.so (called through the
C ABI) against a plain fp32 torch reference of the same op on the same seeded inputs.
bf16 operands are generated in bf16 first, so the only differences are accumulation order
(fp32 in both) and the final bf16 rounding where the output is bf16."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from sais_amd import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


def assert_close(got, ref, atol, rtol=0.0, name=""):
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = (err > tol)
    assert not bad.any(), f"{name}: max err {err.max().item():.3e} (ref max {ref.abs().max().item():.3e}), {bad.sum().item()} bad"


# ------------------------------------------------------------------ MFMA layout: exact small-integer data
def test_gemm_nt_exact_integers(ops):
    from sais_amd import _lib as L
    M, N, K = 200, 256, 128
    g = torch.Generator().manual_seed(1)
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-3, 4, (N, K), generator=g).float()     # asymmetric B
    out = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm_nt(a.to(torch.bfloat16).to(DEV), w.to(torch.bfloat16).to(DEV), L.EPI_BIAS_F32, out)
    assert torch.equal(out.cpu(), a @ w.t())


def test_gemm_tn_exact_integers(ops):
    M, N1, N2 = 333, 256, 128
    g = torch.Generator().manual_seed(2)
    p = torch.randint(-3, 4, (M, N1), generator=g).float()
    q = torch.randint(-3, 4, (M, N2), generator=g).float()
    dW = torch.zeros(N1, N2, device=DEV)
    db = torch.zeros(N1, device=DEV)
    ops.gemm_tn(p.to(torch.bfloat16).to(DEV), q.to(torch.bfloat16).to(DEV), dW, db, nsplit=3)
    assert torch.equal(dW.cpu(), p.t() @ q)
    assert torch.equal(db.cpu(), p.sum(0))


# M >= 8192 takes the persistent eight-wave kernel (ragged last tile, 512-workgroup persistent grid with and without a
# second round); the last-but-one shape is 304 tiles x 9 column tiles = more tiles than persistent workgroups
@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (50432 // 8, 1152, 384), (264, 2048, 384), (128, 384, 1536),
                                   (8192 + 3 * 128 + 57, 384, 384), (50432 - 128 * 90, 1152, 384),
                                   (8192 + 128 + 5, 512, 384), (8192 + 99 * 3 + 7, 1536, 384), (197 * 150 + 77, 384, 384)])
def test_gemm_nt_epilogues(ops, M, N, K):
    from sais_amd import _lib as L
    a = rnd(M, K, seed=3, dtype=torch.bfloat16)
    w = rnd(N, K, seed=4, scale=0.05, dtype=torch.bfloat16)
    bias = rnd(N, seed=5, scale=0.1)
    ref = a.float() @ w.float().t() + bias
    tol = dict(atol=2e-2, rtol=1e-2)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt(a, w, L.EPI_BIAS_BF16, out, bias=bias)
    assert_close(out, ref, name="bias_bf16", **tol)
    ops.gemm_nt(a, w, L.EPI_BIAS_RELU_BF16, out, bias=bias)
    assert_close(out, ref.relu(), name="relu", **tol)
    o32 = torch.empty(M, N, device=DEV)
    ops.gemm_nt(a, w, L.EPI_BIAS_F32, o32, bias=bias)
    assert_close(o32, ref, atol=1e-3, rtol=1e-4, name="f32")
    res = rnd(M, N, seed=6)
    x = res.clone()
    x16 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt(a, w, L.EPI_BIAS_RESID_F32, x, bias=bias, aux=x, out2=x16)       # in place
    assert_close(x, ref + res, atol=1e-3, rtol=1e-4, name="resid")
    assert_close(x16, ref + res, name="resid16", **tol)
    u = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt(a, w, L.EPI_BIAS_GELU_BF16, out, bias=bias, out2=u)
    assert_close(out, F.gelu(ref), name="gelu", **tol)
    assert_close(u, ref, name="gelu_pre", **tol)
    pre = rnd(M, N, seed=7, dtype=torch.bfloat16)
    ops.gemm_nt(a, w, L.EPI_DGELU_BF16, out, aux=pre)
    pf = pre.float().requires_grad_(True)
    F.gelu(pf).sum().backward()
    assert_close(out, (ref - bias) * pf.grad, name="dgelu", **tol)
    ops.gemm_nt(a, w, L.EPI_DRELU_BF16, out, aux=pre)
    assert_close(out, (ref - bias) * (pre.float() > 0), name="drelu", **tol)
    # training pair: forward stores gelu'(pre-activation), backward is a plain multiply
    dg = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt(a, w, L.EPI_BIAS_GELU_GRAD_BF16, out, bias=bias, out2=dg)
    rf = ref.clone().requires_grad_(True)
    F.gelu(rf).sum().backward()
    assert_close(out, F.gelu(ref), name="gelu(+grad)", **tol)
    assert_close(dg, rf.grad, name="gelu_grad", **tol)
    ops.gemm_nt(a, w, L.EPI_MUL_BF16, out, aux=pre)
    assert_close(out, (ref - bias) * pre.float(), name="mul", **tol)
    # the shipped pair (ABI 12): GELU' as one-byte codes q = clamp(rint(26 + 203 d), 0, 255), d = (q - 26) / 203
    dq = torch.empty(M, N, dtype=torch.uint8, device=DEV)
    ops.gemm_nt(a, w, L.EPI_BIAS_GELU_GRADQ_BF16, out, bias=bias, out2=dq)
    assert_close(out, F.gelu(ref), name="gelu(+gradq)", **tol)
    want = (26 + 203 * rf.grad).round().clamp(0, 255)
    assert float((dq.float() - want).abs().max()) <= 1          # one code off only where 26 + 203 d sits on a rounding boundary
    assert float((dq.float() != want).float().mean()) <= 2e-3
    assert float(((dq.float() - 26) / 203 - rf.grad).abs().max()) <= 0.5 / 203 + 1.5e-3      # half a step + the clipped extremes
    codes = torch.randint(0, 256, (M, N), dtype=torch.uint8, generator=torch.Generator().manual_seed(12)).to(DEV)
    ops.gemm_nt(a, w, L.EPI_MULQ_BF16, out, aux=codes)
    assert_close(out, (ref - bias) * ((codes.float() - 26) / 203), name="mulq", **tol)
    with pytest.raises(L.SaisHipError):
        ops.gemm_nt(a, w, L.EPI_BIAS_GELU_GRAD_BF16, out, bias=bias)             # out2 is mandatory
    with pytest.raises(L.SaisHipError):
        ops.gemm_nt(a, w, L.EPI_MULQ_BF16, out)                                  # aux is mandatory
    with pytest.raises(L.SaisHipError):
        ops.gemm_nt(a, w, L.EPI_MUL_BF16, out)                                   # aux is mandatory


def test_gemm_patch_epilogue(ops):
    from sais_amd import _lib as L
    Fr, K, N = 3, 768, 384
    a = rnd(Fr * 196, K, seed=8, dtype=torch.bfloat16)
    w = rnd(N, K, seed=9, scale=0.05, dtype=torch.bfloat16)
    bias, pos = rnd(N, seed=10), rnd(197, N, seed=11)
    tok = torch.zeros(Fr, 197, N, device=DEV)
    ops.gemm_nt(a, w, L.EPI_PATCH_F32, tok, bias=bias, aux=pos, grp=(196, 197, 1))
    ref = (a.float() @ w.float().t() + bias).view(Fr, 196, N) + pos[1:]
    assert_close(tok[:, 1:], ref, atol=1e-3, rtol=1e-4)
    assert tok[:, 0].abs().max().item() == 0.0


@pytest.mark.parametrize("M,N1,N2", [(1000, 256, 128), (50432 // 16, 384, 1536), (264, 1152, 384)])
def test_gemm_tn(ops, M, N1, N2):
    p = rnd(M, N1, seed=12, dtype=torch.bfloat16)
    q = rnd(M, N2, seed=13, dtype=torch.bfloat16)
    dW = torch.zeros(N1, N2, device=DEV)
    db = torch.zeros(N1, device=DEV)
    ops.gemm_tn(p, q, dW, db)
    ref = p.float().t() @ q.float()
    assert_close(dW, ref, atol=2e-3 * math.sqrt(M), rtol=1e-4)
    assert_close(db, p.float().sum(0), atol=1e-3 * math.sqrt(M))
    ops.gemm_tn(p, q, dW, None)                      # accumulates
    assert_close(dW, 2 * ref, atol=4e-3 * math.sqrt(M), rtol=1e-4)


# ------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("rows,eps", [(197 * 3, 1e-6), (1001, 1e-5)])
def test_layernorm_fwd_bwd(ops, rows, eps):
    x = rnd(rows, 384, seed=20, scale=2.0) + 0.3
    gamma, beta = 1 + 0.1 * rnd(384, seed=21), 0.1 * rnd(384, seed=22)
    y16 = torch.empty(rows, 384, dtype=torch.bfloat16, device=DEV)
    y32 = torch.empty(rows, 384, device=DEV)
    mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    ops.layernorm_fwd(x, rows, 384, gamma, beta, eps, y16, y32, mean, rstd)
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (384,), gr, br, eps)
    assert_close(y32, ref, atol=2e-5)
    assert_close(y16, ref, atol=2e-2, rtol=1e-2)
    dy16 = rnd(rows, 384, seed=23, dtype=torch.bfloat16)
    dy32 = rnd(rows, 384, seed=24)
    dres = rnd(rows, 384, seed=25)
    ref.backward(dy16.float() + dy32)
    dx32 = torch.empty(rows, 384, device=DEV)
    dx16 = torch.empty(rows, 384, dtype=torch.bfloat16, device=DEV)
    dg, db = torch.zeros(384, device=DEV), torch.zeros(384, device=DEV)
    ops.layernorm_bwd(x, 384, mean, rstd, gamma, rows, dy16=dy16, dy32=dy32, dres=dres, dx32=dx32, dx16=dx16,
                      dgamma=dg, dbeta=db)
    assert_close(dx32, xr.grad + dres, atol=1e-4, rtol=1e-4)
    assert_close(dx16, xr.grad + dres, atol=3e-2, rtol=1e-2)
    assert_close(dg, gr.grad, atol=2e-3, rtol=1e-4)
    assert_close(db, br.grad, atol=2e-3, rtol=1e-4)


def test_layernorm_cls_rows_strided(ops):
    Fr = 5
    x = rnd(Fr, 197, 384, seed=26)
    gamma, beta = 1 + 0.1 * rnd(384, seed=27), 0.1 * rnd(384, seed=28)
    y = torch.empty(Fr, 384, device=DEV)
    ops.layernorm_fwd(x, Fr, 197 * 384, gamma, beta, 1e-6, y32=y)
    assert_close(y, F.layer_norm(x[:, 0], (384,), gamma, beta, 1e-6), atol=2e-5)


# ------------------------------------------------------------------ ViT attention
def _attn_ref(qkv, frames):
    q, k, v = qkv.float().view(frames, 197, 3, 6, 64).permute(2, 0, 3, 1, 4)
    p = ((q @ k.transpose(-2, -1)) * 0.125).softmax(-1)
    return (p @ v).transpose(1, 2).reshape(frames * 197, 384), p


@pytest.mark.parametrize("frames", [3, 64])
def test_vit_attention_fwd_bwd(ops, frames):
    """frames = 64 -> 384 (frame, head) problems: more than the 256 persistent workgroups of the single-pass backward."""
    qkv = rnd(frames * 197, 1152, seed=30, scale=1.5, dtype=torch.bfloat16)
    out = torch.empty(frames * 197, 384, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(frames, 6, 197, device=DEV)
    probs = torch.empty(frames, 6, 197, 197, device=DEV)
    ops.vit_attn_fwd(qkv, frames, out, lse, probs)
    qr = qkv.float().requires_grad_(True)
    ref, p = _attn_ref(qr, frames)
    assert_close(probs, p, atol=2e-3, rtol=2e-2, name="probs")
    assert_close(out, ref, atol=2e-2, rtol=2e-2, name="out")
    qf, kf = qkv.float().view(frames, 197, 3, 6, 64).permute(2, 0, 3, 1, 4)[:2]
    assert_close(lse, torch.logsumexp((qf @ kf.transpose(-2, -1)) * 0.125, -1), atol=1e-3, name="lse")
    dout = rnd(frames * 197, 384, seed=31, dtype=torch.bfloat16)
    ref.backward(dout.float())
    dqkv = torch.full((frames * 197, 1152), float("nan"), dtype=torch.bfloat16, device=DEV)
    delta = torch.empty(frames, 6, 197, device=DEV)
    ops.vit_attn_bwd(qkv, dout, out, lse, delta, frames, dqkv)
    g = qr.grad
    scale = g.abs().max().item()
    assert_close(dqkv, g, atol=2e-2 * scale, rtol=2e-2, name="dqkv")
    # per-part relative L2 (dQ, dK, dV separately: a swapped or dropped part cannot hide in the tolerance)
    for i, name in enumerate(("dq", "dk", "dv")):
        a, b = dqkv[:, 384 * i:384 * (i + 1)].float(), g[:, 384 * i:384 * (i + 1)]
        assert ((a - b).norm() / b.norm()).item() <= 1.5e-2, name
    # a second call into the same buffers reproduces the result (persistent workgroups, no stale LDS state)
    again = torch.full_like(dqkv, float("nan"))
    ops.vit_attn_bwd(qkv, dout, out, lse, delta, frames, again)
    assert torch.equal(again, dqkv)


@pytest.mark.parametrize("frames,ntok", [(5, 197), (70, 197), (9, 37)])
def test_vit_attention_cls_query_only(ops, frames, ntok):
    """The last block's attention restricted to the CLS query (sais_vit_attn_cls_fwd / _bwd) vs fp32 torch autograd of the
    full attention with the output gradient zero outside the CLS rows: out on [frames, 384]; dqkv complete (dk, dv of every
    token, dq on the CLS rows, exact zeros elsewhere)."""
    qkv = rnd(frames * ntok, 1152, seed=33, scale=1.5, dtype=torch.bfloat16)
    out_c = torch.full((frames, 384), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.vit_attn_cls_fwd(qkv, frames, out_c, ntok)
    qr = qkv.float().requires_grad_(True)
    t = qr.view(frames, ntok, 3, 6, 64).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax((t[0] @ t[1].transpose(-2, -1)) * 0.125, -1) @ t[2]).transpose(1, 2).reshape(frames, ntok, 384)
    assert_close(out_c, ref[:, 0], atol=2e-2, rtol=2e-2, name="out (CLS rows)")
    dout_c = rnd(frames, 384, seed=34, dtype=torch.bfloat16)
    ref[:, 0].backward(dout_c.float())
    dqkv = torch.full((frames * ntok, 1152), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.vit_attn_cls_bwd(qkv, dout_c, frames, dqkv, ntok)
    g = qr.grad
    assert torch.isfinite(dqkv.float()).all()
    for i, name in enumerate(("dq", "dk", "dv")):
        a, b = dqkv[:, 384 * i:384 * (i + 1)].float(), g[:, 384 * i:384 * (i + 1)]
        assert ((a - b).norm() / b.norm()).item() <= 6e-3, name              # bf16 rounding of the outputs only
    dq = dqkv[:, :384].float().view(frames, ntok, 384)
    assert float(dq[:, 1:].abs().max()) == 0.0 and float(dq[:, 0].abs().max()) > 0
    if ntok == 197:                                       # and against the full kernels on the same data
        out = torch.empty(frames * 197, 384, dtype=torch.bfloat16, device=DEV)
        ops.vit_attn_fwd(qkv, frames, out)
        assert_close(out_c, out.view(frames, 197, 384)[:, 0], atol=2e-2, rtol=2e-2, name="vs the full forward")


def test_vit_attention_forced_peaky_rows(ops):
    """A spiked key per query forces near-one-hot softmax rows (exercises the max subtraction)."""
    frames = 1
    qkv = rnd(197, 1152, seed=32, scale=0.5, dtype=torch.bfloat16)
    qkv[:, 384:768] *= 12.0
    out = torch.empty(197, 384, dtype=torch.bfloat16, device=DEV)
    ops.vit_attn_fwd(qkv, frames, out)
    ref, _ = _attn_ref(qkv, frames)
    assert torch.isfinite(out.float()).all()
    assert_close(out, ref, atol=3e-2, rtol=3e-2)


# ------------------------------------------------------------------ embedding glue, optimizer
def test_patchify_matches_conv_unfold(ops):
    Fr = 2
    img = rnd(Fr, 3, 224, 224, seed=40)
    patches = torch.empty(Fr * 196, 768, dtype=torch.bfloat16, device=DEV)
    ops.patchify(img, patches)
    ref = img.reshape(Fr, 3, 14, 16, 14, 16).permute(0, 2, 4, 1, 3, 5).reshape(Fr * 196, 768)
    assert torch.equal(patches.cpu(), ref.to(torch.bfloat16).cpu())


def test_cls_rows_and_embed_bwd(ops):
    Fr = 4
    cls, pos = rnd(384, seed=41), rnd(197, 384, seed=42)
    tok = torch.zeros(Fr, 197, 384, device=DEV)
    ops.vit_cls_rows(cls, pos, tok, Fr)
    assert_close(tok[:, 0], (cls + pos[0]).expand(Fr, -1), atol=0)
    dtok = rnd(Fr, 197, 384, seed=43)
    dcls, dpos = torch.zeros(384, device=DEV), torch.zeros(197, 384, device=DEV)
    dpatch = torch.empty(Fr * 196, 384, dtype=torch.bfloat16, device=DEV)
    ops.vit_embed_bwd(dtok, Fr, dcls, dpos, dpatch)
    assert_close(dcls, dtok[:, 0].sum(0), atol=1e-5)
    assert_close(dpos, dtok.sum(0), atol=1e-5)
    assert torch.equal(dpatch.cpu(), dtok[:, 1:].reshape(-1, 384).to(torch.bfloat16).cpu())


def test_sgd_cast_transpose(ops):
    n = 384 * 1152 + 3
    p, g = rnd(n, seed=44), rnd(n, seed=45)
    ref = p - 0.1 * 0.5 * g
    sh = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    ops.sgd_step(p, g, sh, 0.1, 0.5)
    assert_close(p, ref, atol=1e-6)                      # lr*scale*g is FMA-contracted on the device
    assert torch.equal(sh.cpu(), p.to(torch.bfloat16).cpu())
    w = rnd(1152, 384, seed=46)
    wt = torch.empty(384, 1152, dtype=torch.bfloat16, device=DEV)
    ops.transpose_cast_bf16(w, 1152, 384, wt)
    assert torch.equal(wt.cpu(), w.t().contiguous().to(torch.bfloat16).cpu())
    c = torch.empty(1152 * 384, dtype=torch.bfloat16, device=DEV)
    ops.cast_bf16(w, c)
    assert torch.equal(c.view(1152, 384).cpu(), w.to(torch.bfloat16).cpu())


def test_transpose_batch(ops):
    """Every shadow of a model in one launch: ragged shapes (not multiples of 32), bf16 and f32 destinations."""
    shapes = [(1152, 384), (384, 384), (37, 65), (1, 384), (256, 384), (33, 31)]
    srcs = [rnd(r, c, seed=300 + i) for i, (r, c) in enumerate(shapes)]
    for dt in (torch.bfloat16, torch.float32):
        dsts = [torch.full((c, r), 7.0, dtype=dt, device=DEV) for (r, c) in shapes]
        table, n, tiles = ops.transpose_table(list(zip(srcs, dsts)), DEV)
        assert n == len(shapes) and tiles == sum(((r + 31) // 32) * ((c + 31) // 32) for r, c in shapes)
        ops.transpose_batch(table, n, tiles, dt == torch.float32)
        for s, d in zip(srcs, dsts):
            assert torch.equal(d.cpu(), s.t().contiguous().to(dt).cpu())


# ------------------------------------------------------------------ temporal glue + head + loss
def test_temporal_prepare(ops):
    B, T = 3, 9
    x = rnd(B, 1, T, 384, seed=50)
    pos, cls = rnd(T, 384, seed=51), rnd(384, seed=52)
    z32 = torch.empty(B * (T + 1), 384, device=DEV)
    z16 = torch.empty(B * (T + 1), 384, dtype=torch.bfloat16, device=DEV)
    ops.temporal_prepare_fwd(x, T * 384, 384, pos, cls, B, T, z32, z16)
    ref = torch.cat((cls.expand(B, 1, 384), x[:, 0] + pos), 1)
    assert_close(z32.view(B, T + 1, 384), ref, atol=0)
    dz32 = rnd(B * (T + 1), 384, seed=53)
    slabs = rnd(3, B * (T + 1), 384, seed=54)             # raw split-K slabs of the in_proj dX GEMM, summed on load
    for use_add, use_slabs in ((True, True), (True, False), (False, True)):
        dx = torch.empty(B, 1, T, 384, device=DEV)
        dpos, dcls = torch.zeros(T, 384, device=DEV), torch.zeros(384, device=DEV)
        ops.temporal_prepare_bwd(dz32 if use_add else None, slabs if use_slabs else None, B, T, dx, T * 384, 384, False,
                                 dpos, dcls)
        d = ((dz32 if use_add else 0) + (slabs.sum(0) if use_slabs else 0)).view(B, T + 1, 384)
        assert_close(dx[:, 0], d[:, 1:], atol=1e-5)
        assert_close(dpos, d[:, 1:].sum(0), atol=1e-4)
        assert_close(dcls, d[:, 0].sum(0), atol=1e-4)


@pytest.mark.parametrize("S,lens", [(33, [32, 20, 3, 0, 17]), (16, [15, 15]), (64, [63, 10, 40]), (40, [39, 7]), (96, [95, 50, 1]),
                                    (8, [7, 3, 0])])
def test_temporal_attention_fwd_bwd(ops, S, lens):
    B = len(lens)
    qkv = rnd(B * S, 1152, seed=60, scale=1.0)
    pad = torch.zeros(B, S, dtype=torch.bool)
    for b, n in enumerate(lens):
        pad[b, n + 1:] = True
    padu = pad.to(torch.uint8).to(DEV)
    ctx = torch.empty(B * S, 384, device=DEV)
    avg = torch.empty(B, S, S, device=DEV)
    ops.temporal_attn_fwd(qkv, padu, B, S, ctx, avg)
    qr = qkv.clone().requires_grad_(True)
    q, k, v = qr.view(B, S, 3, 4, 96).permute(2, 0, 3, 1, 4)
    sc = (q * 96 ** -0.5) @ k.transpose(-2, -1)
    sc = sc.masked_fill(pad.to(DEV).view(B, 1, 1, S), float("-inf"))
    p = sc.softmax(-1)
    ref = (p @ v).transpose(1, 2).reshape(B * S, 384)
    assert_close(avg, p.mean(1), atol=1e-5, name="attn_avg")
    assert_close(ctx, ref, atol=1e-4, name="ctx")
    dctx = rnd(B * S, 384, seed=61)
    ref.backward(dctx)
    dqkv = torch.empty(B * S, 1152, device=DEV)
    ops.temporal_attn_bwd(qkv, padu, B, S, dctx, dqkv)
    assert_close(dqkv, qr.grad, atol=1e-4 * max(1.0, qr.grad.abs().max().item()), name="dqkv")
    # dctx handed over as raw split-K slabs (summed on load)
    parts = rnd(3, B * S, 384, seed=62)
    parts[2] = dctx - parts[0] - parts[1]
    dqkv2 = torch.empty_like(dqkv)
    ops.temporal_attn_bwd(qkv, padu, B, S, parts, dqkv2)
    assert_close(dqkv2, qr.grad, atol=2e-4 * max(1.0, qr.grad.abs().max().item()), name="dqkv (slabs)")


@pytest.mark.parametrize("M,N,K", [(264, 1152, 384), (48, 384, 2048), (300, 256, 128)])
def test_gemm_nt_f32_bf16x3(ops, M, N, K):
    """fp32-operand GEMM (hi/lo bf16 split, 3 MFMA products): ~fp32 accuracy, all four epilogues."""
    from sais_amd import _lib as L
    a, w = rnd(M, K, seed=90), rnd(N, K, seed=91, scale=0.05)
    bias, aux = rnd(N, seed=92, scale=0.1), rnd(M, N, seed=93)
    ref = (a.double() @ w.double().t()).float() + bias
    tol = 2e-5 * math.sqrt(K)
    out = torch.empty(M, N, device=DEV)
    ops.gemm_nt_f32(a, w, L.EPI_BIAS_F32, out, bias=bias)
    assert_close(out, ref, atol=tol, name="bias")
    ops.gemm_nt_f32(a, w, L.EPI_BIAS_RELU_F32, out, bias=bias)
    assert_close(out, ref.relu(), atol=tol, name="relu")
    ops.gemm_nt_f32(a, w, L.EPI_BIAS_RESID_F32, out, bias=bias, aux=aux)
    assert_close(out, ref + aux, atol=tol, name="resid")
    ops.gemm_nt_f32(a, w, L.EPI_DRELU_F32, out, aux=aux)
    assert_close(out, (ref - bias) * (aux > 0), atol=tol, name="drelu")
    wt = torch.empty(K, N, device=DEV)
    ops.transpose_f32(w, N, K, wt)
    assert torch.equal(wt, w.t().contiguous())
    p32, q32 = rnd(M, 256, seed=94), rnd(M, 128, seed=95)
    dW = torch.zeros(256, 128, device=DEV)
    ops.gemm_tn(p32, q32, dW, None)
    assert_close(dW, p32.to(torch.bfloat16).float().t() @ q32.to(torch.bfloat16).float(), atol=2e-3 * math.sqrt(M))


@pytest.mark.parametrize("M,N,K", [(264, 1152, 384), (264, 384, 2048), (264, 2048, 384), (48, 384, 1152), (301, 64, 128)])
def test_tgemm_bf16x3_epilogues_and_slabs(ops, M, N, K):
    """sais_tgemm (the temporal encoder's 64 x 64 bf16x3 GEMM): every epilogue at ~fp32 accuracy, raw split-K slabs that sum
    to the product for every legal split, dropout masks = the exported masks of the site."""
    from sais_amd import _lib as L
    a, w = rnd(M, K, seed=190), rnd(N, K, seed=191, scale=0.05)
    bias, aux = rnd(N, seed=192, scale=0.1), rnd(M, N, seed=193)
    prod = (a.double() @ w.double().t()).float()
    tol = 2e-5 * math.sqrt(K)
    out = torch.empty(M, N, device=DEV)
    ops.tgemm(a, w, L.TG_BIAS, out, bias=bias)
    assert_close(out, prod + bias, atol=tol, name="bias")
    ops.tgemm(a, w, L.TG_BIAS, out)
    assert_close(out, prod, atol=tol, name="no bias")
    ops.tgemm(a, w, L.TG_BIAS_RELU, out, bias=bias)
    assert_close(out, (prod + bias).relu(), atol=tol, name="relu")
    ops.tgemm(a, w, L.TG_DRELU, out, aux=aux)
    assert_close(out, prod * (aux > 0), atol=tol, name="drelu")
    nk = K // 64
    for ns in [d for d in (1, 2, 3, 4, 6, 8, 16) if nk % d == 0]:
        slabs = torch.full((ns, M, N), float("nan"), device=DEV)
        ops.tgemm(a, w, L.TG_RAW, slabs, nsplit=ns)
        assert_close(slabs.sum(0), prod, atol=tol, name=f"raw slabs, nsplit {ns}")
    with pytest.raises(L.SaisHipError):
        ops.tgemm(a, w, L.TG_BIAS, out, nsplit=2)                      # only raw partial sums can be split
    # train-mode dropout in the epilogues: mask element m * N + n of the site
    st = ops.rng_state(3, DEV)
    keep = ops.dropout_mask(M * N, 0.25, st, 5, DEV).view(M, N).float() / 0.75
    ops.tgemm(a, w, L.TG_BIAS_RELU, out, bias=bias, drop=(0.25, st, 5))
    assert_close(out, (prod + bias).relu() * keep, atol=2 * tol, name="relu + dropout")
    ops.tgemm(a, w, L.TG_DRELU, out, aux=aux, drop=(0.25, st, 5))
    assert_close(out, prod * (aux > 0) * keep, atol=2 * tol, name="drelu + dropout")


@pytest.mark.parametrize("M,nslab", [(264, 3), (13, 1), (1000, 8)])
def test_temporal_ln_fwd_bwd_consume_slabs(ops, M, nslab):
    """y = resid + drop(sum slabs + bias), z = LayerNorm(y) and its backward with dy = sum slabs + add, vs torch."""
    slabs = rnd(nslab, M, 384, seed=300)
    bias, resid = rnd(384, seed=301, scale=0.1), rnd(M, 384, seed=302, scale=2.0)
    gamma, beta = 1 + 0.1 * rnd(384, seed=303), 0.05 * rnd(384, seed=304)
    st = ops.rng_state(9, DEV)
    for p_drop in (0.0, 0.1):
        keep = 1.0 if p_drop == 0 else ops.dropout_mask(M * 384, p_drop, st, 2, DEV).view(M, 384).float() / (1 - p_drop)
        drop = None if p_drop == 0 else (p_drop, st, 2)
        y, z = torch.empty(M, 384, device=DEV), torch.empty(M, 384, device=DEV)
        mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
        ops.temporal_ln_fwd(slabs, bias, resid, gamma, beta, 1e-5, z, y=y, mean=mean, rstd=rstd, drop=drop)
        y_ref = resid + (slabs.sum(0) + bias) * keep
        assert_close(y, y_ref, atol=1e-5, name="y")
        assert_close(z, F.layer_norm(y_ref, (384,), gamma, beta, 1e-5), atol=2e-5, name="z")
        assert_close(mean, y_ref.mean(1), atol=1e-5, name="mean")
        assert_close(rstd, 1 / torch.sqrt(y_ref.var(1, unbiased=False) + 1e-5), atol=0, rtol=1e-5, name="rstd")
        z2 = torch.empty_like(z)
        ops.temporal_ln_fwd(slabs, bias, resid, gamma, beta, 1e-5, z2, drop=drop)          # inference form: nothing saved
        assert torch.equal(z2, z)
        # backward at (y, mean, rstd)
        add = rnd(M, 384, seed=305)
        dy_ref = slabs.sum(0) + add
        yr = y_ref.clone().requires_grad_(True)
        gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        F.layer_norm(yr, (384,), gr, br, 1e-5).backward(dy_ref)
        dx, dxd = torch.empty(M, 384, device=DEV), torch.empty(M, 384, device=DEV)
        dg, db = torch.zeros(384, device=DEV), torch.zeros(384, device=DEV)
        ops.temporal_ln_bwd(slabs, add, y, mean, rstd, gamma, dx, dx_drop=None if drop is None else dxd, drop=drop,
                            dgamma=dg, dbeta=db)
        sc = max(1.0, dy_ref.abs().max().item())
        assert_close(dx, yr.grad, atol=2e-5 * sc, name="dx")
        if drop is not None:
            assert_close(dxd, yr.grad * keep, atol=4e-5 * sc, name="dropout(dx)")
        assert_close(dg, gr.grad, atol=2e-5 * sc * math.sqrt(M), name="dgamma")
        assert_close(db, br.grad, atol=2e-5 * sc * math.sqrt(M), name="dbeta")
        dx2 = torch.empty_like(dx)
        ops.temporal_ln_bwd(None, dy_ref, y, mean, rstd, gamma, dx2)                          # add only, no column sums
        assert_close(dx2, yr.grad, atol=2e-5 * sc, name="dx (add only)")


@pytest.mark.parametrize("two_stream", [False, True])
def test_head_fwd_bwd(ops, two_stream):
    B, S = 5, 7
    zr, zf = rnd(B, S, 384, seed=70), rnd(B, S, 384, seed=71)
    W, bias = rnd(256, 384, seed=72, scale=0.05), rnd(256, seed=73, scale=0.1)
    rep, emb = torch.empty(B, 384, device=DEV), torch.empty(B, 256, device=DEV)
    ops.head_fwd(zr, zf if two_stream else None, S * 384, B, W, bias, rep, emb)
    zr_, zf_ = zr.clone().requires_grad_(True), zf.clone().requires_grad_(True)
    W_, b_ = W.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    r = F.relu(zr_)[:, 0] + (F.relu(zf_)[:, 0] if two_stream else 0)
    ref = F.linear(F.relu(r), W_, b_)
    assert_close(emb, ref, atol=1e-4)
    demb = rnd(B, 256, seed=74)
    ref.backward(demb)
    dW, db = torch.zeros(256, 384, device=DEV), torch.zeros(256, device=DEV)
    dzr, dzf = torch.zeros(B, S, 384, device=DEV), torch.zeros(B, S, 384, device=DEV)
    ops.head_bwd(demb, W, rep, zr, zf if two_stream else None, S * 384, B, dW, db, dzr, dzf if two_stream else None)
    assert_close(dW, W_.grad, atol=1e-4)
    assert_close(db, b_.grad, atol=1e-5)
    assert_close(dzr, zr_.grad, atol=1e-5)
    if two_stream:
        assert_close(dzf, zf_.grad, atol=1e-5)


@pytest.mark.parametrize("B,C", [(8, 2), (5, 3), (1, 2)])
def test_nce_loss_and_grads(ops, B, C):
    from oracle import sais_oracle as O
    emb = rnd(B, 256, seed=80)
    protos = torch.rand(C, 256, generator=torch.Generator().manual_seed(81)).to(DEV)
    lab = torch.randint(0, C, (B,), generator=torch.Generator().manual_seed(82))
    sim, probs = torch.empty(B, C, device=DEV), torch.empty(B, C, device=DEV)
    loss = torch.empty(1, device=DEV)
    demb, dpro = torch.empty(B, 256, device=DEV), torch.zeros(C, 256, device=DEV)
    ops.nce(emb, protos, lab.int().to(DEV), sim, probs, loss, demb, dpro)
    e = emb.cpu().requires_grad_(True)
    pd = {str(c): protos[c:c + 1].cpu().clone().requires_grad_(True) for c in range(C)}
    ref = O.nce_loss(e, lab, pd)
    ref.backward()
    assert_close(sim, O.cosine_logits(e, pd).detach(), atol=1e-6)
    assert_close(probs, O.probs_from_logits(O.cosine_logits(e, pd)).detach(), atol=1e-6)
    assert abs(loss.item() - ref.item()) < 1e-6
    assert_close(demb, e.grad, atol=1e-7)
    assert_close(dpro, torch.cat([pd[str(c)].grad for c in range(C)]), atol=1e-7)


@pytest.mark.parametrize("M", [197 * 16, 197 * 64, 197 * 64 + 32, 197 * 256, 197 * 64 + 64])
def test_gemm_tn_grouped_matches_individual(ops, M):
    """The four weight-gradient GEMMs of a ViT block in one launch == four separate launches.  M = 12 608, 50 432 and
    12 672 (whole numbers of 64-row steps, >= 8192; even and odd step counts per split) take the wide 128x384 kernel,
    the other two the 128x128 one."""
    shapes = [(384, 1536), (1536, 384), (384, 384), (1152, 384)]
    items, refs = [], []
    for i, (n1, n2) in enumerate(shapes):
        p = rnd(M, n1, seed=100 + i, dtype=torch.bfloat16)
        q = rnd(M, n2, seed=110 + i, dtype=torch.bfloat16)
        dW, db = torch.zeros(n1, n2, device=DEV), torch.zeros(n1, device=DEV)
        items.append((p, q, dW, db if i != 2 else None))
        refs.append((p.float().t() @ q.float(), p.float().sum(0)))
    ops.gemm_tn_grouped(items, M)
    for (p, q, dW, db), (rw, rb) in zip(items, refs):
        assert_close(dW, rw, atol=2e-3 * math.sqrt(M), rtol=1e-4)
        if db is not None:
            assert_close(db, rb, atol=1e-3 * math.sqrt(M))


def test_gemm_tn_grouped_large_tile_exact_integers_and_bitwise_repeatable(ops):
    """The 192 x 384 dW kernel (gemm_tn_xl.hip: one wave per SIMD, hand-counted vmcnt / lgkmcnt, LDS-DMA ring): small-integer
    operands make every product and sum exact, so dW and db must EQUAL the fp64 result (a swapped fragment, a stale LDS stage
    or a missed wait shows as a wrong integer), asymmetric P / Q catch a transposed store, and the slab + finish form has no
    atomics, so forty launches on the same inputs must be bit-identical (a race screen: LDS-DMA data that arrives late is rare
    and load-dependent).  M = 197 x 96 = 18 912 rows = 591 steps of 32: splits of 59 and 60 steps (both loop parities)."""
    M = 197 * 96
    g = torch.Generator().manual_seed(11)
    shapes = [(384, 1536), (1536, 384), (384, 384), (1152, 384)]
    ps = [torch.randint(-2, 3, (M, n1), generator=g).float() for n1, _ in shapes]
    qs = [torch.randint(-3, 4, (M, n2), generator=g).float() for _, n2 in shapes]
    def run():
        items = [(p.to(torch.bfloat16).to(DEV), q.to(torch.bfloat16).to(DEV), torch.zeros(p.shape[1], q.shape[1], device=DEV),
                  torch.zeros(p.shape[1], device=DEV) if i != 2 else None) for i, (p, q) in enumerate(zip(ps, qs))]
        ops.gemm_tn_grouped(items, M)
        return items
    first = run()
    for (p, q), (_, _, dW, db) in zip(zip(ps, qs), first):
        assert torch.equal(dW.cpu().double(), p.double().t() @ q.double())
        if db is not None:
            assert torch.equal(db.cpu().double(), p.double().sum(0))
    # busy neighbours: a second stream streams through HBM while the launches repeat
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, device=DEV)
    for rep in range(40):
        with torch.cuda.stream(side):
            junk.add_(1.0)
        again = run()
        for (_, _, dW, db), (_, _, dW2, db2) in zip(first, again):
            assert torch.equal(dW, dW2), f"repetition {rep}: dW differs"
            assert db is None or torch.equal(db, db2), f"repetition {rep}: db differs"
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,shapes", [
    (8192, [(384, 1536), (1536, 384), (384, 384), (1152, 384)]),            # smallest M of the regime: 256 steps, 10 splits of 25-26: falls back
    (197 * 128, [(384, 1536), (1536, 384), (384, 384), (1152, 384)]),       # 788 steps: 10 splits of 78-79
    (197 * 512, [(384, 1536), (1536, 384)]),                                # config 4's M = 100 864: 16 tiles x 16 splits of 197
    (197 * 256, [(768, 384)]),                                              # the pruned block's k/v-only launch: 4 tiles, 64 splits < 48 steps: falls back
    (197 * 256, [(384, 384), (384, 768), (768, 1152)]),                     # uneven tile counts per item: 2 + 4 + 12 = 18 tiles x 14 splits
    (197 * 64, [(384, 1536), (1536, 384), (384, 384), (1152, 384)] * 6 + [(768, 384)]),   # R6.8: six blocks + the k / v item = 148 tiles, ONE
])                                                                          # split: every tile has one owner (plain read-add-write)
def test_gemm_tn_grouped_large_tile_strided_views_and_accumulation(ops, M, shapes):
    """The large-tile dW kernel over what its callers hand it: P / Q as column slices of wider buffers (leading dimension > N, base
    pointer 16-B but not 128-B aligned), dW as a slice of a wider gradient buffer, a SECOND call that must add to the first (+=), db
    present or not per item — and the launches at the edges of its regime, which must fall back to the 128 x 384 kernel and give the
    same numbers.  Small-integer operands: every sum is exact in fp32 in any order, so the results must EQUAL the reference."""
    g = torch.Generator().manual_seed(M % 1000 + len(shapes))
    items, refs = [], []
    for i, (n1, n2) in enumerate(shapes):
        pw = torch.randint(-2, 3, (M, n1 + 24), generator=g).to(torch.bfloat16).to(DEV)
        qw = torch.randint(-3, 4, (M, n2 + 40), generator=g).to(torch.bfloat16).to(DEV)
        p, q = pw[:, 8:8 + n1], qw[:, 16:16 + n2]
        wide = torch.randint(-5, 6, (n1, n2 + 8), generator=g).float().to(DEV)
        dW = wide[:, 4:4 + n2]
        db = torch.randint(-5, 6, (n1,), generator=g).float().to(DEV) if i % 2 == 0 else None
        refs.append((dW.clone() + 2 * (p.float().t() @ q.float()), None if db is None else db + 2 * p.float().sum(0), wide.clone()))
        items.append((p, q, dW, db))
    ops.gemm_tn_grouped(items, M)
    ops.gemm_tn_grouped(items, M)
    for (p, q, dW, db), (rw, rb, wide0) in zip(items, refs):
        assert torch.equal(dW, rw)
        assert db is None or torch.equal(db, rb)
        wide = dW._base if dW._base is not None else dW
        assert torch.equal(wide[:, :4], wide0[:, :4]) and torch.equal(wide[:, -4:], wide0[:, -4:]), "columns beside the dW slice were written"


_SLAB_SCRIPT = r"""
import math, sys, torch
sys.path.insert(0, sys.argv[1])
from sais_amd import ops
M = 197 * 64
g = torch.Generator().manual_seed(3)
mk = lambda n: (torch.randn(M, n, generator=g)).to(torch.bfloat16).cuda()
shapes = [(384, 1536), (1536, 384), (384, 384), (1152, 384)]
ops_in = [(mk(a), mk(b)) for a, b in shapes]
def run(times):
    items = [(p, q, torch.zeros(p.shape[1], q.shape[1], device="cuda"), torch.zeros(p.shape[1], device="cuda")) for p, q in ops_in]
    for _ in range(times):
        ops.gemm_tn_grouped(items, M)
    torch.cuda.synchronize()
    return items
a, b, twice = run(1), run(1), run(2)
for (p, q, dW, db), (_, _, dW2, db2), (_, _, dW3, db3) in zip(a, b, twice):
    assert torch.equal(dW, dW2) and torch.equal(db, db2), "slab mode must be bit-reproducible"
    ref = p.float().t() @ q.float()
    assert (dW - ref).abs().max().item() <= 2e-3 * math.sqrt(M) + 1e-4 * ref.abs().max().item()
    assert (db - p.float().sum(0)).abs().max().item() <= 1e-3 * math.sqrt(M)
    assert torch.allclose(dW3, 2 * dW, rtol=1e-6, atol=1e-4) and torch.allclose(db3, 2 * db, rtol=1e-6, atol=1e-4)   # accumulates (+=)
print("SLAB_OK")
"""


def test_gemm_tn_grouped_slab_mode_deterministic_and_accumulating(ops):
    """SAIS_TN_SLABS=1 (sais_gemm_tn_grouped_ws, ABI 10): the M-splits of the wide dW kernel write raw slabs and a fixed-order
    finish adds them to dW / db: two runs are bit-identical, a second call accumulates.  The switch is read once per
    process, so the check runs in a child."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _SLAB_SCRIPT, root], env=dict(os.environ, SAIS_TN_SLABS="1"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SLAB_OK" in r.stdout, r.stdout + r.stderr


# ------------------------------------------------------------------ row-owning GEMM with LayerNorm in the epilogue
def _ln_ref(x, gamma, beta, eps):
    return F.layer_norm(x, (384,), gamma, beta, eps)


# M >= 28 672 takes the eight-wave tile (two 112-row halves, one workgroup per CU): ragged (197 x 146 + 5 rows -> 113-row
# tiles: the second half stores ONE row) and whole-frame tiles
@pytest.mark.parametrize("M,K", [(300, 384), (8192 + 128 + 57, 1536), (12608, 384), (197 * 146 + 5, 1536), (197 * 200, 384)])
def test_gemm_ln_fwd(ops, M, K):
    """x_out = A.W^T + b + resid ; xn = LayerNorm(x_out) in ONE launch vs fp32 torch (ragged last tile included)."""
    a = rnd(M, K, seed=200, dtype=torch.bfloat16)
    w = rnd(384, K, seed=201, scale=0.05, dtype=torch.bfloat16)
    bias, resid = rnd(384, seed=202, scale=0.1), rnd(M, 384, seed=203, scale=2.0)
    resid[:, 7] += 30.0                                   # an outlier channel: mean/variance must not lose it
    gamma, beta = 1 + 0.1 * rnd(384, seed=204), 0.05 * rnd(384, seed=205)
    x_out = torch.empty(M, 384, device=DEV)
    xn = torch.empty(M, 384, dtype=torch.bfloat16, device=DEV)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    ops.gemm_ln_fwd(a, w, bias, resid, x_out, xn, gamma, beta, 1e-6, mean, rstd)
    ref_x = a.float() @ w.float().t() + bias + resid
    assert_close(x_out, ref_x, atol=2e-3, rtol=1e-5, name="x_out")
    assert_close(mean, ref_x.mean(1), atol=1e-4, name="mean")
    assert_close(rstd, 1.0 / torch.sqrt(ref_x.var(1, unbiased=False) + 1e-6), atol=0, rtol=1e-4, name="rstd")
    assert_close(xn, _ln_ref(x_out, gamma, beta, 1e-6), atol=2e-2, rtol=1e-2, name="xn")      # bf16 rounding of the output
    # in place on the residual stream (the inference path)
    r2 = resid.clone()
    ops.gemm_ln_fwd(a, w, bias, r2, r2, xn, gamma, beta, 1e-6)
    assert torch.equal(r2, x_out)


@pytest.mark.parametrize("M,K", [(300, 1152), (8192 + 57, 1536), (12608, 1152), (197 * 146 + 5, 1152), (197 * 200, 1536)])
def test_gemm_ln_bwd(ops, M, K):
    """dy = A.W^T ; dx = dres + dLN(dy) ; dgamma, dbeta — vs torch autograd of layer_norm on the fp32 dy."""
    a = rnd(M, K, seed=210, scale=0.5, dtype=torch.bfloat16)
    w = rnd(384, K, seed=211, scale=0.05, dtype=torch.bfloat16)
    x = rnd(M, 384, seed=212, scale=1.5)
    x[:, 11] -= 20.0
    gamma = 1 + 0.1 * rnd(384, seed=213)
    dres = rnd(M, 384, seed=214)
    mean = x.mean(1)
    rstd = 1.0 / torch.sqrt(x.var(1, unbiased=False) + 1e-6)
    dgamma, dbeta = torch.zeros(384, device=DEV), torch.zeros(384, device=DEV)
    dx32 = dres.clone()                                    # in place, as the ViT backward uses it
    dx16 = torch.empty(M, 384, dtype=torch.bfloat16, device=DEV)
    ops.gemm_ln_bwd(a, w, x, mean, rstd, gamma, dres=dx32, dx32=dx32, dx16=dx16, dgamma=dgamma, dbeta=dbeta)
    dy = a.float() @ w.float().t()
    xr = x.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    br = torch.zeros(384, device=DEV, requires_grad=True)
    F.layer_norm(xr, (384,), gr, br, 1e-6).backward(dy)
    scale = dy.abs().max().item()
    assert_close(dx32, xr.grad + dres, atol=2e-4 * scale, rtol=1e-4, name="dx32")
    assert_close(dx16, xr.grad + dres, atol=2e-2, rtol=1e-2, name="dx16")
    assert_close(dgamma, gr.grad, atol=2e-4 * scale * math.sqrt(M), rtol=1e-4, name="dgamma")
    assert_close(dbeta, br.grad, atol=2e-4 * scale * math.sqrt(M), rtol=1e-4, name="dbeta")
    # without the residual gradient / without the column sums
    dx_only = torch.empty(M, 384, device=DEV)
    ops.gemm_ln_bwd(a, w, x, mean, rstd, gamma, dx32=dx_only)
    assert_close(dx_only, xr.grad, atol=2e-4 * scale, rtol=1e-4, name="dx (no dres)")


@pytest.mark.parametrize("M,K,case", [(197 * 64, 1536, "plain"), (197 * 146 + 5, 1152, "plain"), (197 * 64, 1536, "outlier"),
                                      (197 * 64, 1536, "tiny_gamma")])
def test_gemm_ln_bwd_from_the_saved_bf16_layernorm_output(ops, M, K, case):
    """ABI 11: with the forward's saved bf16 LayerNorm output y the epilogue rebuilds xhat = (y - beta) / gamma instead of reading the
    fp32 LayerNorm input (half the bytes).  vs torch autograd on the fp32 input: dx within 1e-3 of max|dy| (xhat enters dx only
    through xhat * mean(dy g xhat)), dgamma within 4e-3 relative L2 (measured 1-2e-3: y's bf16 rounding averages out over the rows; 8e-3 for 'outlier', measured 2.6e-3).  'outlier':
    gamma in [0.2, 3], beta up to +-2 (|beta / gamma| up to 10: the rounding of y is amplified).  'tiny_gamma': one gamma = 1e-5 ->
    every workgroup must fall back to the fp32 input (exact result, as without y)."""
    a = rnd(M, K, seed=220, scale=0.5, dtype=torch.bfloat16)
    w = rnd(384, K, seed=221, scale=0.05, dtype=torch.bfloat16)
    x = rnd(M, 384, seed=222, scale=1.5)
    x[:, 11] -= 20.0
    gamma, beta = 1 + 0.1 * rnd(384, seed=223), 0.05 * rnd(384, seed=224)
    if case == "outlier":
        gamma = (0.2 + 2.8 * torch.rand(384, generator=torch.Generator().manual_seed(5))).to(DEV)
        beta = 2.0 * rnd(384, seed=225).clamp(-1, 1)
    if case == "tiny_gamma":
        gamma[77] = 1e-5
    dres = rnd(M, 384, seed=226)
    mean = x.mean(1)
    rstd = 1.0 / torch.sqrt(x.var(1, unbiased=False) + 1e-6)
    y16 = F.layer_norm(x, (384,), gamma, beta, 1e-6).to(torch.bfloat16)
    dgamma, dbeta = torch.zeros(384, device=DEV), torch.zeros(384, device=DEV)
    dx32 = dres.clone()
    dx16 = torch.empty(M, 384, dtype=torch.bfloat16, device=DEV)
    ops.gemm_ln_bwd(a, w, x, mean, rstd, gamma, dres=dx32, dx32=dx32, dx16=dx16, dgamma=dgamma, dbeta=dbeta, xn16=y16, beta=beta)
    dy = a.float() @ w.float().t()
    xr, gr = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True)
    br = beta.clone().requires_grad_(True)
    F.layer_norm(xr, (384,), gr, br, 1e-6).backward(dy)
    scale = dy.abs().max().item()
    exact = case == "tiny_gamma"
    assert_close(dx32, xr.grad + dres, atol=(2e-4 if exact else 1e-3) * scale * max(1.0, gamma.abs().max().item()), rtol=1e-4, name="dx32")
    rel = lambda got, ref: float((got - ref).norm() / ref.norm())
    gbar = 1e-4 if exact else (8e-3 if case == "outlier" else 4e-3)         # |beta / gamma| up to 10 amplifies y's bf16 rounding
    assert rel(dgamma, gr.grad) <= gbar, rel(dgamma, gr.grad)
    assert_close(dbeta, br.grad, atol=2e-4 * scale * math.sqrt(M), rtol=1e-4, name="dbeta")
    from parity import parity_log
    parity_log(f"ln_bwd from bf16 y[{case}]/dgamma rel-L2", rel(dgamma, gr.grad), gbar)
    parity_log(f"ln_bwd from bf16 y[{case}]/dx max-abs / max|dy|", float((dx32 - xr.grad - dres).abs().max()) / scale, 1e-3 * max(1.0, gamma.abs().max().item()))


def _gelu_parts(u):
    cdf = 0.5 * (1.0 + torch.erf(u / math.sqrt(2.0)))
    return u * cdf, cdf + u * torch.exp(-0.5 * u * u) / math.sqrt(2.0 * math.pi)


@pytest.mark.parametrize("M,ln,dp", [(8192 + 57, True, False), (197 * 64, True, True), (197 * 146 + 5, False, True),
                                     (197 * 256, True, False)])
def test_mlp_fused_fwd(ops, M, ln, dp):
    """sais_mlp_fwd (fc1 + GELU / GELU' -> fc2 + residual -> LayerNorm in ONE launch) vs (a) fp32 torch on the same bf16
    operands and (b) the two-launch form it replaces (gemm_nt<GELU_GRAD> + gemm_ln_fwd / gemm_nt<RESID>), ragged last tile,
    DropPath row scales, with and without the following LayerNorm, training (h, g written) and inference (nothing
    materialised) forms."""
    from sais_amd import _lib as L
    H = 1536
    xn2 = rnd(M, 384, seed=300, dtype=torch.bfloat16)
    w1 = rnd(H, 384, seed=301, scale=0.06, dtype=torch.bfloat16)
    w2 = rnd(384, H, seed=302, scale=0.04, dtype=torch.bfloat16)
    b1, b2 = rnd(H, seed=303, scale=0.3), rnd(384, seed=304, scale=0.1)
    resid = rnd(M, 384, seed=305, scale=2.0)
    resid[:, 7] += 30.0
    gamma, beta = 1 + 0.1 * rnd(384, seed=306), 0.05 * rnd(384, seed=307)
    rs = None
    if dp:
        rs = (torch.rand(M, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5)) > 0.2).float() / 0.8
    x_out = torch.empty(M, 384, device=DEV)
    xn = torch.empty(M, 384, dtype=torch.bfloat16, device=DEV) if ln else None
    mean, rstd = (torch.empty(M, device=DEV), torch.empty(M, device=DEV)) if ln else (None, None)
    h = torch.full((M, H), 7.0, dtype=torch.bfloat16, device=DEV)
    g = torch.full((M, H), 7.0, dtype=torch.bfloat16, device=DEV)
    ops.mlp_fwd(xn2, w1, b1, w2, b2, resid, x_out, h=h, g=g, xn_out=xn, gamma=gamma if ln else None,
                beta=beta if ln else None, eps=1e-6, mean=mean, rstd=rstd, rowscale=rs)
    # (a) fp32 reference; the second GEMM sees the bf16-rounded h, as in the kernel
    u = xn2.float() @ w1.float().t() + b1
    h_ref, g_ref = _gelu_parts(u)
    assert_close(h, h_ref, atol=2e-2, rtol=1e-2, name="h")
    assert_close(g, g_ref, atol=2e-2, rtol=1e-2, name="g")
    y = h.float() @ w2.float().t() + b2
    ref_x = resid + (y * rs[:, None] if dp else y)
    assert_close(x_out, ref_x, atol=3e-3, rtol=1e-5, name="x_out")
    if ln:
        assert_close(mean, ref_x.mean(1), atol=1e-4, name="mean")
        assert_close(rstd, 1.0 / torch.sqrt(ref_x.var(1, unbiased=False) + 1e-6), atol=0, rtol=1e-4, name="rstd")
        assert_close(xn, _ln_ref(x_out, gamma, beta, 1e-6), atol=2e-2, rtol=1e-2, name="xn")
    # (b) the two-launch form: same bf16 h / g (same arithmetic, element for element), same x_out up to fp32 summation order
    h2, g2 = torch.empty_like(h), torch.empty_like(g)
    ops.gemm_nt(xn2, w1, L.EPI_BIAS_GELU_GRAD_BF16, h2, bias=b1, out2=g2)
    assert torch.equal(h2, h) and torch.equal(g2, g)
    x2 = torch.empty_like(x_out)
    if ln:
        xn_2 = torch.empty_like(xn)
        ops.gemm_ln_fwd(h2, w2, b2, resid, x2, xn_2, gamma, beta, 1e-6, rowscale=rs)
    else:
        ops.gemm_nt(h2, w2, L.EPI_BIAS_RESID_F32, x2, bias=b2, aux=resid, rowscale=rs)
    assert_close(x_out, x2, atol=1e-5, rtol=1e-6, name="x_out vs two launches")
    # inference form: nothing materialised, in place on the residual stream
    r2 = resid.clone()
    ops.mlp_fwd(xn2, w1, b1, w2, b2, r2, r2, xn_out=xn, gamma=gamma if ln else None, beta=beta if ln else None,
                eps=1e-6, rowscale=rs)
    assert_close(r2, x_out, atol=1e-5, rtol=1e-6, name="in place, h not materialised")


@pytest.mark.parametrize("M,dp", [(8192 + 57, False), (197 * 146 + 5, True), (197 * 256, False)])
def test_mlp_fused_bwd(ops, M, dp):
    """sais_mlp_bwd (dX fc2 x GELU' -> dX fc1 -> LayerNorm backward in ONE launch) vs the two launches it replaces
    (gemm_nt<MUL> + gemm_ln_bwd) and vs fp32 torch."""
    from sais_amd import _lib as L
    H = 1536
    d16 = rnd(M, 384, seed=320, scale=0.5, dtype=torch.bfloat16)
    w2t = rnd(H, 384, seed=321, scale=0.05, dtype=torch.bfloat16)          # fc2.weight^T
    w1t = rnd(384, H, seed=322, scale=0.05, dtype=torch.bfloat16)          # fc1.weight^T
    g = (0.5 + 0.6 * rnd(M, H, seed=323)).to(torch.bfloat16)
    x = rnd(M, 384, seed=324, scale=1.5)
    x[:, 11] -= 20.0
    gamma = 1 + 0.1 * rnd(384, seed=325)
    dres = rnd(M, 384, seed=326)
    mean = x.mean(1)
    rstd = 1.0 / torch.sqrt(x.var(1, unbiased=False) + 1e-6)
    rs = None
    if dp:
        rs = (torch.rand(M, device=DEV, generator=torch.Generator(device=DEV).manual_seed(6)) > 0.2).float() / 0.8
    dgamma, dbeta = torch.zeros(384, device=DEV), torch.zeros(384, device=DEV)
    dx32 = dres.clone()
    dx16 = torch.empty(M, 384, dtype=torch.bfloat16, device=DEV)
    du = torch.full((M, H), 7.0, dtype=torch.bfloat16, device=DEV)
    ops.mlp_bwd(d16, w2t, g, w1t, du, x, mean, rstd, gamma, dres=dx32, dx32=dx32, dx16=dx16, dgamma=dgamma, dbeta=dbeta,
                rowscale16=rs)
    # the two-launch form
    du2 = torch.empty_like(du)
    ops.gemm_nt(d16, w2t, L.EPI_MUL_BF16, du2, aux=g)
    assert torch.equal(du2, du)
    dg2, db2 = torch.zeros(384, device=DEV), torch.zeros(384, device=DEV)
    dx32_2, dx16_2 = dres.clone(), torch.empty_like(dx16)
    ops.gemm_ln_bwd(du2, w1t, x, mean, rstd, gamma, dres=dx32_2, dx32=dx32_2, dx16=dx16_2, dgamma=dg2, dbeta=db2, rowscale16=rs)
    scale = float((du.float() @ w1t.float().t()).abs().max())
    assert_close(dx32, dx32_2, atol=1e-5 * scale, rtol=1e-5, name="dx32 vs two launches")
    assert_close(dgamma, dg2, atol=2e-5 * scale * math.sqrt(M), rtol=1e-4, name="dgamma vs two launches")
    assert_close(dbeta, db2, atol=2e-5 * scale * math.sqrt(M), rtol=1e-4, name="dbeta vs two launches")
    # fp32 torch
    dy = du.float() @ w1t.float().t()
    xr = x.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    br = torch.zeros(384, device=DEV, requires_grad=True)
    F.layer_norm(xr, (384,), gr, br, 1e-6).backward(dy)
    assert_close(du, (d16.float() @ w2t.float().t()) * g.float(), atol=2e-2, rtol=1e-2, name="du")
    assert_close(dx32, xr.grad + dres, atol=2e-4 * scale, rtol=1e-4, name="dx32")
    want16 = (xr.grad + dres) * (rs[:, None] if dp else 1.0)
    assert_close(dx16, want16, atol=2e-2, rtol=1e-2, name="dx16")
    assert_close(dgamma, gr.grad, atol=2e-4 * scale * math.sqrt(M), rtol=1e-4, name="dgamma")
    assert_close(dbeta, br.grad, atol=2e-4 * scale * math.sqrt(M), rtol=1e-4, name="dbeta")


@pytest.mark.parametrize("M,N,K,ks", [(256, 384, 1536, 12), (200, 384, 384, 6), (37, 1536, 384, 4), (256, 384, 1536, 5)])
def test_gemm_nt_splitk_small_m(ops, M, N, K, ks):
    """Deterministic split-K for the [frames, 384] GEMMs of the CLS-only last block: raw slabs + sais_splitk_finish (bias,
    row scale, fp32 residual, f32 and bf16 outputs) vs fp32 torch; ragged M, a split that does not divide K / 64."""
    a = rnd(M, K, seed=70, dtype=torch.bfloat16)
    w = rnd(N, K, seed=71, scale=0.05, dtype=torch.bfloat16)
    bias, rs, aux = rnd(N, seed=72, scale=0.2), 0.5 + torch.rand(M, device=DEV), rnd(M, N, seed=73)
    o32 = torch.full((M, N), float("nan"), device=DEV)
    o16 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt_splitk(a, w, ks, bias=bias, rowscale=rs, aux=aux, out32=o32, out16=o16)
    ref = aux + rs[:, None] * (a.float() @ w.float().t() + bias)
    assert_close(o32, ref, atol=2e-3, rtol=1e-5, name="out32")
    assert_close(o16, ref, atol=2e-2, rtol=1e-2, name="out16")
    again = torch.empty_like(o32)
    ops.gemm_nt_splitk(a, w, ks, bias=bias, rowscale=rs, aux=aux, out32=again)
    assert torch.equal(again, o32)                               # fixed summation order
    plain = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt_splitk(a, w, ks, out16=plain)
    assert_close(plain, a.float() @ w.float().t(), atol=2e-2, rtol=1e-2, name="plain")


def test_integration_md_ctypes_stub_runs_as_written(ops):
    """The ctypes binding shown in INTEGRATION.md (what a SAIS maintainer would paste) is executed verbatim: the struct
    layout in the document must match include/sais_hip.h, and the call must give fc1 + GELU."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(import ctypes, torch.*?)```", md, re.S).group(1)
    code = code.replace('ctypes.CDLL("sais_amd/libsais_hip.so")', f'ctypes.CDLL("{os.path.join(root, "sais_amd", "libsais_hip.so")}")')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    x = rnd(300, 384, seed=1, dtype=torch.bfloat16)
    w = rnd(1536, 384, seed=2, scale=0.05, dtype=torch.bfloat16)
    b = rnd(1536, seed=3, scale=0.1)
    out = ns["linear_gelu"](x, w, b)
    ref = F.gelu(x.float() @ w.float().t() + b)
    assert_close(out.float(), ref, atol=2e-2, rtol=2e-2)


@pytest.mark.parametrize("nsplit", [1, 2])
def test_gemm_tn_grouped_f32_accumulates_owned_or_atomic(ops, nsplit):
    """The temporal encoder's dW launch (fp32 operands, M = a few hundred rows): one M-split = every workgroup owns its
    tile (plain read-add-write), more = atomics; both must ADD to what dW / db already hold."""
    M = 264
    shapes = [(384, 2048), (2048, 384), (384, 384), (1152, 384)]
    items, refs = [], []
    for i, (n1, n2) in enumerate(shapes):
        p, q = rnd(M, n1, seed=300 + i), rnd(M, n2, seed=310 + i)
        dW, db = rnd(n1, n2, seed=320 + i), rnd(n1, seed=330 + i)
        refs.append((dW.clone() + p.bfloat16().float().t() @ q.bfloat16().float(), db.clone() + p.bfloat16().float().sum(0)))
        items.append((p, q, dW, db))
    ops.gemm_tn_grouped(items, M, nsplit=nsplit)
    for (p, q, dW, db), (rw, rb) in zip(items, refs):
        assert_close(dW, rw, atol=2e-3 * math.sqrt(M), rtol=1e-4)
        assert_close(db, rb, atol=1e-3 * math.sqrt(M))


# ------------------------------------------------------------------ the alternate shipped forms of the dW launch stay CORRECT
@pytest.mark.parametrize("switch,value", [
    ("SAIS_TN_XL", "0"),              # dW on the 128 x 384 ping-pong kernel + atomics (rounds 2-5)
    ("SAIS_TN_XL_SLABS", "0")])       # dW on the 192 x 384 kernel with fp32 atomics (LABNOTES R6.1)
def test_alternate_dw_forms_pass_the_same_tests(ops, switch, value):
    """The switches are read once per process: the same parity tests as the default form, in a child process with the switch set.
    (The forms that were measured and REJECTED live behind -DSAIS_EXPERIMENTAL: tests/test_experimental.py.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sel = "gemm_tn and not alternate and not slab_mode and not bitwise_repeatable"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_kernels_gpu.py"), "-q", "-x", "-k", sel],
                       env=dict(os.environ, **{switch: value}), cwd=root, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-1500:] + r.stderr[-500:]


def test_bf16_gelu_grad_form_passes_the_model_gradient_tests(ops):
    """SAIS_GELU_GRAD_Q8=0 (read once per process): GELU' saved as bf16 (epilogues 10 / 11, rounds 3-5) instead of one-byte codes —
    the ViT gradient tests against the golden vectors in a child process with the switch set, block API and per-launch path."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_model_gpu.py"), "-q", "-x", "-k",
                        "vit_grads or e2e_train_step or block_level or pruned_last_block"],
                       env=dict(os.environ, SAIS_GELU_GRAD_Q8="0"), cwd=root, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-1500:] + r.stderr[-500:]
