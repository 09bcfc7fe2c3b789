"""Train-mode dropout of the temporal encoder on a real MI355X.

The reference trains with nn.TransformerEncoderLayer's default dropout = 0.1 (prepare_model.py:75, model.train() at
train.py:59).  The masks here come from Philox inside the HIP kernels (this library's stream, not torch's), so parity is
checked by exporting the masks a forward used (fullModel.dropout_masks) and handing them to the oracle, whose train-mode
arithmetic is itself pinned against the reference's (tests/test_dropout_oracle.py, dropout.npz)."""
import math
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import synth  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"
LOGIT_TOL, ATTN_TOL, GRAD_REL = 1e-3, 2e-3, 4e-2


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from sais_amd import ops as o
    return o


def rel_l2(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / max(float(b.norm()), 1e-12))


def test_dropout_kernel_masks(ops):
    n, p = 1 << 20, 0.1
    st = ops.rng_state(1234, DEV)
    x = torch.randn(n, device=DEV)
    r = torch.randn(n, device=DEV)
    m = ops.dropout_mask(n, p, st, 5, DEV).bool()
    out = ops.dropout(x, p, st, 5, resid=r, out=torch.empty_like(x))
    inv = (torch.tensor(1.0) / (torch.tensor(1.0) - torch.tensor(p))).item()   # fp32 1 / (1 - p), as the kernel computes it
    want = r + torch.where(m, x * inv, torch.zeros_like(x))
    assert torch.equal(out, want)
    keep = m.float().mean().item()
    assert abs(keep - (1 - p)) < 4 * math.sqrt(p * (1 - p) / n), keep          # Bernoulli(0.9): 4 sigma
    # neighbouring elements / sites / offsets are independent draws: agreement rate ~ 0.9^2 + 0.1^2 = 0.82
    m_site = ops.dropout_mask(n, p, st, 6, DEV).bool()
    assert abs((m == m_site).float().mean().item() - 0.82) < 0.01
    assert abs((m[1:] == m[:-1]).float().mean().item() - 0.82) < 0.01
    st2 = st.clone()
    ops.rng_advance(st2)
    assert st2.tolist() == [1234, 1]
    assert abs((m == ops.dropout_mask(n, p, st2, 5, DEV).bool()).float().mean().item() - 0.82) < 0.01
    assert torch.equal(m, ops.dropout_mask(n, p, st.clone(), 5, DEV).bool())   # same state, same mask
    other_seed = ops.rng_state(1235, DEV)
    assert abs((m == ops.dropout_mask(n, p, other_seed, 5, DEV).bool()).float().mean().item() - 0.82) < 0.01
    # in place, no residual; p = 0 keeps everything
    y = x.clone()
    ops.dropout(y, p, st, 5)
    assert torch.equal(y, torch.where(m, x * inv, torch.zeros_like(x)))
    assert bool(ops.dropout_mask(4096, 0.0, st, 0, DEV).all())


def _model(modal):
    from sais_amd.temporal import fullModel
    m = fullModel('reps', 2, 'in_vs_out', 384, 'ViT', modalities=modal)
    m.load_state_dict(synth.temporal_state_dict(seed=1), strict=True)
    return m.to(DEV)


def _inputs(lens, T, seeds=(910, 911)):
    B = len(lens)
    x, f = synth.reps(seed=seeds[0], B=B, T=T), synth.reps(seed=seeds[1], B=B, T=T)
    for b, n in enumerate(lens):
        x[b, :, n:] = 0
        f[b, :, n:] = 0
    return x, f, synth.padding_mask(lens)


@pytest.mark.parametrize("modal", ["RGB", "RGB-Flow"])
def test_train_mode_forward_backward_vs_oracle_with_the_same_masks(ops, modal):
    from oracle import sais_oracle as O
    from sais_amd.loss import calcNCELoss, cosine_logits_and_probs
    lens, T, C = [9, 4, 7, 9], 9, 2
    B, S = len(lens), T + 1
    m = _model(modal).train()
    m.dropout_seed = 77
    x, f, pad = _inputs(lens, T)
    xg = x.to(DEV).requires_grad_(True)
    protos = torch.nn.ParameterDict({k: torch.nn.Parameter(v.clone().to(DEV)) for k, v in synth.prototypes(2, C).items()})
    lab = synth.labels(seed=912, B=B, nclasses=C)
    fd = f.to(DEV) if modal == "RGB-Flow" else None
    emb, attn = m(xg, fd, lens, lens, 'Prototypes', pad.to(DEV), pad.to(DEV) if fd is not None else None, None)
    loss = calcNCELoss(0, emb, lab, [f"v{b}" for b in range(B)], protos, None)
    loss.backward()
    st = m.last_dropout_state
    assert st.tolist() == [77, 1]
    drop = {"rgb": [{k: v.cpu() for k, v in lm.items()} for lm in m.dropout_masks(st, B, S, stream=0)]}
    if modal == "RGB-Flow":
        drop["flow"] = [{k: v.cpu() for k, v in lm.items()} for lm in m.dropout_masks(st, B, S, stream=1)]
    rate = torch.cat([v.flatten().float() for lm in drop["rgb"] for v in lm.values()]).mean().item()
    assert abs(rate - 0.9) < 0.005, rate

    sd = {k: v.clone().requires_grad_(True) for k, v in synth.temporal_state_dict(seed=1).items()}
    pr = {k: v.clone().requires_grad_(True) for k, v in synth.prototypes(2, C).items()}
    xr = x.clone().requires_grad_(True)
    e_ref, a_ref = O.temporal_forward(sd, xr, f if modal == "RGB-Flow" else None, pad, pad, modal, drop=drop, p=m.dropout_p)
    l_ref = O.nce_loss(e_ref, lab, pr)
    l_ref.backward()
    sim, _ = cosine_logits_and_probs(emb, protos)
    assert (sim.cpu() - O.cosine_logits(e_ref, pr).detach()).abs().max().item() <= LOGIT_TOL
    assert abs(loss.item() - l_ref.item()) <= LOGIT_TOL
    assert (attn.cpu() - a_ref.detach()).abs().max().item() <= ATTN_TOL       # the dropped map (torch 1.8 returns that one)
    assert rel_l2(xg.grad, xr.grad) <= GRAD_REL
    P = dict(m.named_parameters())
    bad = {}
    for n in ("linear.weight", "linear.bias", "frame_cls", "frame_pos_embeddings.0", "frame_pos_embeddings.8"):
        bad[n] = rel_l2(P[n].grad, sd[n].grad)
    for l in range(4):
        for t in ("self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight", "self_attn.out_proj.bias",
                  "linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias", "norm1.weight", "norm1.bias",
                  "norm2.weight", "norm2.bias"):
            n = f"transEncoderFrame.layers.{l}.{t}"
            bad[n] = rel_l2(P[n].grad, sd[n].grad)
    for k in protos.keys():
        bad["proto" + k] = rel_l2(protos[k].grad, pr[k].grad)
    bad = {k: v for k, v in bad.items() if v > GRAD_REL}
    assert not bad, bad


def test_eval_is_identity_and_train_draws_fresh_masks(ops):
    lens, T = [9, 4, 7, 9], 9
    x, f, pad = _inputs(lens, T)
    m = _model("RGB")
    args = (x.to(DEV), None, lens, lens, 'Prototypes', pad.to(DEV), None, None)
    with torch.no_grad():
        e_eval, _ = m.eval()(*args)
        m.dropout_p = 0.0
        e_p0, _ = m.train()(*args)
        assert torch.equal(e_eval, e_p0) and m._rng is None       # eval / p = 0: the dropout-free kernels, no RNG touched
        m.dropout_p = 0.1
        e1, _ = m(*args)
        e2, _ = m(*args)
    assert m._rng.tolist() == [0, 2]
    assert (e1 - e_eval).abs().max().item() > 0.05 and (e1 - e2).abs().max().item() > 0.05


def test_graph_replays_draw_fresh_masks_and_match_eager(ops):
    from sais_amd.graph import GraphedStep
    from sais_amd.loss import calcNCELoss, label_columns
    from sais_amd.optim import SGD
    lens, T, B = [9, 9, 9, 9], 9, 4
    x, _, pad = _inputs(lens, T)
    xd, padd = x.to(DEV), pad.to(DEV)
    m = _model("RGB").train()
    protos = torch.nn.ParameterDict({k: torch.nn.Parameter(v.clone().to(DEV)) for k, v in synth.prototypes(2, 2).items()})
    cols = label_columns(synth.labels(seed=912, B=B), protos, DEV)
    opt = SGD(list(m.parameters()) + list(protos.values()), lr=0.0, engines=[m])      # lr 0: weights stay put
    names = [f"v{b}" for b in range(B)]

    def step():
        opt.zero_grad()
        emb, _ = m(xd, None, lens, lens, 'Prototypes', padd, None, None)
        loss = calcNCELoss(0, emb, cols, names, protos, None)
        loss.backward()
        opt.step()
        return loss

    graphed = GraphedStep(step, warmup=2)
    l1 = float(graphed())
    state_after_1 = m._rng.clone()
    l2 = float(graphed())
    assert l1 != l2 and m._rng.tolist()[1] == state_after_1.tolist()[1] + 1    # the advance is a graph node
    # the same RNG state through the eager path gives the replay's loss and gradients
    g_graph = m.flat.grad.clone()
    m._rng.copy_(state_after_1)
    le = float(step())
    assert le == l2
    assert rel_l2(m.flat.grad, g_graph) <= 1e-5


@pytest.mark.parametrize("train", [True, False])
def test_layer_level_c_calls_equal_the_per_launch_sequence(ops, train):
    """sais_temporal_layer_fwd / _bwd (one C call per encoder layer: SURVEY 8b) vs the per-launch Python sequence: the same
    launches in the same order — embeddings and loss bit-identical, the head-averaged attention map up to the order of its four
    atomic head additions, gradients equal (the layer's dW launch is owner-computes), in train mode with dropout (same Philox
    state) and in eval()."""
    import sais_amd.temporal as TM
    from sais_amd.loss import calcNCELoss
    lens, T = [9, 4, 7, 9], 9
    x, f, pad = _inputs(lens, T)
    lab = synth.labels(seed=912, B=4)

    def run(layer_calls):
        old = TM._LAYER_CALLS
        TM._LAYER_CALLS = layer_calls
        try:
            m = _model("RGB-Flow")
            m = m.train() if train else m.eval()
            m.dropout_seed = 31
            xg, fg = x.to(DEV).requires_grad_(True), f.to(DEV).requires_grad_(True)
            protos = torch.nn.ParameterDict({k: torch.nn.Parameter(v.clone().to(DEV)) for k, v in synth.prototypes(2, 2).items()})
            emb, attn = m(xg, fg, lens, lens, 'Prototypes', pad.to(DEV), pad.to(DEV), None)
            loss = calcNCELoss(0, emb, lab, [f"v{b}" for b in range(4)], protos, None)
            loss.backward()
            return emb.detach(), attn.detach(), float(loss), m.flat.grad.clone(), xg.grad.clone(), fg.grad.clone()
        finally:
            TM._LAYER_CALLS = old

    a, b = run(True), run(False)
    assert torch.equal(a[0], b[0]) and a[2] == b[2]
    assert float((a[1] - b[1]).abs().max()) <= 1e-6
    assert float(b[3].abs().max()) > 0
    for u, v in zip(a[3:], b[3:]):
        assert float((u - v).norm() / v.norm()) <= 1e-6


# ------------------------------------------------------------------ DropPath (stochastic depth) of the ViT blocks
def _vit(depth, rate):
    from sais_amd.vit import vit_small
    v = vit_small(patch_size=16, drop_path_rate=rate, depth=depth)
    v.load_state_dict(synth.vit_state_dict(seed=0, depth=depth), strict=True)
    return v.to(DEV)


# M = 1182: stand-alone kernels; M = 9456: LN-fused row GEMMs (four-wave tile); M = 28 762: the eight-wave tile
@pytest.mark.parametrize("frames,depth", [(6, 4), (48, 2), (146, 2)])
def test_droppath_train_mode_vs_oracle_with_the_same_draws(ops, frames, depth):
    """vision_transformer.py:27-46,105-113 in train(): per-sample keep / (1 - p_i) on both residual branches, forward and
    backward.  The draws are this library's (Philox): the per-row scales a forward used are exported, reduced to per-sample
    factors and handed to the oracle, whose DropPath arithmetic is pinned against the reference (droppath.npz)."""
    from oracle import sais_oracle as O
    rate = 0.3
    vit = _vit(depth, rate).train()
    vit.drop_path_seed = 11
    x = synth.clips(seed=931, B=1, T=frames)[0]
    w = synth.reps(seed=932, B=1, T=frames)[0, 0]
    feat = vit(x.to(DEV))
    (feat * w.to(DEV)).sum().backward()
    sc = vit.last_droppath_scales                                       # [2 * depth, frames * 197]
    assert tuple(sc.shape) == (2 * depth, frames * 197)
    per = sc.view(2 * depth, frames, 197)
    assert bool((per == per[:, :, :1]).all())                           # one draw per sample, shared by its 197 rows
    fac = per[:, :, 0].cpu()
    rates = torch.linspace(0, rate, depth).repeat_interleave(2)
    for j in range(2 * depth):
        vals = set(round(float(v), 5) for v in fac[j].unique())
        assert vals <= {0.0, round(1.0 / (1.0 - float(rates[j])), 5)}, (j, vals)
    assert bool((fac[:2] == 1).all()) and float((fac == 0).sum()) >= 1  # block 0: p = 0; something was dropped
    sd = {k: v.clone().requires_grad_(True) for k, v in synth.vit_state_dict(seed=0, depth=depth).items()}
    ref = O.vit_forward(sd, x, depth=depth, droppath=fac)
    (ref * w).sum().backward()
    assert (feat.detach().cpu() - ref.detach()).abs().max().item() <= 3e-2 * ref.detach().abs().max().item()
    bad = {n: rel_l2(q.grad, sd[n].grad) for n, q in vit.named_parameters()}
    bad = {k: v for k, v in bad.items() if v > 6e-2}
    assert not bad, bad
    # a dropped branch sends nothing into its parameters' gradients from that sample: with every sample dropped on one
    # branch the branch's weights get exactly zero — checked through the scales themselves: all-zero rows exist only by
    # chance, so check the forward instead: eval() ignores the rate
    with torch.no_grad():
        e1 = vit.eval()(x.to(DEV))
        e0 = _vit(depth, 0.0).eval()(x.to(DEV))
    assert torch.equal(e1, e0) and (feat.detach() - e1).abs().max().item() > 0.05


def test_droppath_draws_are_fresh_per_forward_and_bernoulli(ops):
    from sais_amd.vit import vit_small
    rates = torch.tensor([0.0, 0.0, 0.25, 0.25, 0.5, 0.5], device=DEV)
    st = ops.rng_state(5, DEV)
    a = ops.droppath_scales(rates, 4096, 3, st)
    assert tuple(a.shape) == (6, 4096 * 3)
    keep = (a.view(6, 4096, 3)[:, :, 0] > 0).float().mean(1).cpu()
    for j, p in enumerate([0.0, 0.0, 0.25, 0.25, 0.5, 0.5]):
        assert abs(float(keep[j]) - (1 - p)) < 4 * math.sqrt(max(p * (1 - p), 1e-9) / 4096) + 1e-9, (j, float(keep[j]))
    assert not torch.equal(a[2], a[3])                                  # branches are independent draws
    ops.rng_advance(st)
    assert not torch.equal(a, ops.droppath_scales(rates, 4096, 3, st))
