"""Module-level parity on a real MI355X against the committed golden vectors (produced by the
reference itself, tests/golden/make_golden.py) and against the pinned CPU oracle.

Tolerances (stated here, used below):
  * class logits (cosine similarities)              <= 1e-3 max-abs   (north-star bar)
  * returned temporal attention maps                <= 2e-3 max-abs
  * embeddings / ViT features (bf16 MFMA operands)  <= 2e-2 * max|ref|
  * gradients (bf16 operands, fp32 accumulation)    <= 4e-2 relative L2 per tensor
"""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"

LOGIT_TOL, ATTN_TOL, FEAT_REL, GRAD_REL = 1e-3, 2e-3, 2e-2, 4e-2      # FEAT_REL: ONE bar with tests/test_bench_size_gpu.py (worst seen 1.7e-2, outlier fixture)


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return True


def maxabs(a, b):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    return float(np.abs(a - np.asarray(b)).max())


def rel_l2(a, b):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = np.asarray(b)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-12))


def make_vit(depth=12):
    from sais_amd.vit import vit_small
    m = vit_small(patch_size=16, drop_path_rate=0.1, depth=depth)
    m.load_state_dict(synth.vit_state_dict(seed=0, depth=depth), strict=True)
    return m.to(DEV).eval()


def make_full(nclasses=2, modalities="RGB-Flow", nlayers=4):
    from sais_amd.temporal import fullModel
    m = fullModel('reps', nclasses, 'in_vs_out', 384, 'ViT', modalities=modalities)
    if nlayers != 4:
        m.transEncoderFrame.layers = m.transEncoderFrame.layers[:nlayers]
        m.transEncoderClip.layers = m.transEncoderClip.layers[:nlayers]
    m.load_state_dict(synth.temporal_state_dict(seed=1, nlayers=nlayers), strict=True)
    m.dropout_p = 0.0                      # the goldens of these tests are dropout-free; tests/test_dropout_gpu.py covers p > 0
    return m.to(DEV).eval()


def protos_dev(C):
    return torch.nn.ParameterDict({k: torch.nn.Parameter(v.clone().to(DEV)) for k, v in synth.prototypes(2, C).items()})


# ------------------------------------------------------------------ ViT
def test_vit_forward_vs_golden(gpu, golden):
    g = golden("vit")
    vit = make_vit()
    x = synth.clips(seed=10, B=1, T=2)[0].to(DEV)
    with torch.no_grad():
        rep = vit(x)
        attn = vit.get_last_selfattention(x)
    ref = g["rep"]
    assert maxabs(rep, ref) <= FEAT_REL * np.abs(ref).max(), maxabs(rep, ref)
    assert maxabs(attn[:, :, [0, 57, 196], :], g["attn_rows"]) <= ATTN_TOL
    assert maxabs(attn.sum(dim=2), g["attn_colsum"]) <= 2e-2
    # cosine logits of the features against 3 random prototypes: the quantity the 1e-3 bar is about
    p = torch.randn(3, 384, generator=torch.Generator().manual_seed(5))
    def cos(a):
        a = torch.as_tensor(a).float().cpu()
        return (a / a.norm(dim=1, keepdim=True)) @ (p / p.norm(dim=1, keepdim=True)).t()
    assert maxabs(cos(rep), cos(ref).numpy()) <= LOGIT_TOL


def test_vit_is_deterministic_and_batch_invariant(gpu):
    vit = make_vit(depth=2)
    x = synth.clips(seed=11, B=1, T=5)[0].to(DEV)
    with torch.no_grad():
        a, b = vit(x), vit(x)
        c = vit(x[1:3])
    assert torch.equal(a, b)
    assert torch.equal(a[1:3], c)


def test_vit_grads_vs_golden(gpu, golden):
    g = golden("vit")
    vit = make_vit()                       # eval(), as the reference ran for the goldens: DropPath 0.1 is the identity there
    x = synth.clips(seed=10, B=1, T=2)[0].to(DEV)
    rep = vit(x)
    (rep * torch.from_numpy(g["grad_wvec"]).to(DEV)).sum().backward()
    worst = {}
    for name, p in vit.named_parameters():
        ref = g["grad/" + name]
        gr = p.grad
        assert gr is not None, name
        if gr.dim() <= 1 or name == "cls_token":
            got = gr
        elif name == "pos_embed":
            got = gr[:, list(g["rows"])]
        else:
            got = gr[:8]
        worst[name] = rel_l2(got, ref)
        nrm = float(g["gnorm/" + name])
        assert abs(gr.norm().item() - nrm) <= GRAD_REL * nrm, (name, gr.norm().item(), nrm)
    bad = {k: v for k, v in worst.items() if v > GRAD_REL}
    assert not bad, bad


# ------------------------------------------------------------------ temporal encoder + head
def _case_inputs(lens):
    T, B = max(lens), len(lens)
    x = synth.reps(seed=100 + T, B=B, T=T)
    f = synth.reps(seed=200 + T, B=B, T=T)
    for b, n in enumerate(lens):
        x[b, :, n:] = 0
        f[b, :, n:] = 0
    return x.to(DEV), f.to(DEV), synth.padding_mask(lens).to(DEV)


@pytest.mark.parametrize("modal", ["RGB", "RGB-Flow"])
def test_temporal_forward_vs_golden(gpu, golden, modal):
    g = golden("temporal")
    m = make_full(2, modal)
    for cname in ("T15", "T12r", "T9r", "T32r"):
        key = f"{modal}/{cname}/"
        lens = [int(v) for v in g[key + "lens"]]
        x, f, pad = _case_inputs(lens)
        x0 = x.clone()
        with torch.no_grad():
            emb, attn = m(x, f, lens, lens, 'Prototypes', pad, pad, None)
        assert torch.equal(x, x0), "inputs must not be mutated"
        ref = g[key + "emb"]
        assert maxabs(emb, ref) <= FEAT_REL * np.abs(ref).max(), (cname, maxabs(emb, ref))
        assert maxabs(attn, g[key + "attn"]) <= ATTN_TOL, (cname, maxabs(attn, g[key + "attn"]))
        protos = protos_dev(2)
        from sais_amd.loss import cosine_logits_and_probs
        from oracle import sais_oracle as O
        sim, probs = cosine_logits_and_probs(emb, protos)
        sim_ref = O.cosine_logits(torch.from_numpy(ref), synth.prototypes(2, 2))
        assert maxabs(sim, sim_ref.numpy()) <= LOGIT_TOL, (cname, maxabs(sim, sim_ref.numpy()))


def test_temporal_tta_list_path(gpu, golden):
    g = golden("temporal")
    m = make_full(2, "RGB-Flow")
    xs, fs, pads, lens_l = [], [], [], []
    for v, T in enumerate((15, 12, 9)):
        xs.append(synth.reps(seed=300 + v, B=2, T=T).to(DEV))
        fs.append(synth.reps(seed=400 + v, B=2, T=T).to(DEV))
        pads.append(synth.padding_mask([T, T]).to(DEV))
        lens_l.append([T, T])
    with torch.no_grad():
        embs, attn = m(xs, fs, lens_l, lens_l, 'Prototypes', pads, pads, None)
    for v in range(3):
        ref = g[f"TTA/emb{v}"]
        assert maxabs(embs[v], ref) <= FEAT_REL * np.abs(ref).max()
    assert maxabs(attn, g["TTA/attn"]) <= ATTN_TOL


@pytest.mark.parametrize("mod", ["RGB-Flow", "RGB", "Flow"])
def test_tta_merged_pass_equals_per_version_passes(gpu, mod):
    """Inference runs the three TTA versions of both streams as ONE encoder pass (stacked sequences, padded under the key mask):
    embeddings and the returned attention map must equal the per-version passes (ragged lengths inside a version included)."""
    from sais_amd import temporal as tmod
    m = make_full(2, mod).eval()
    xs, fs, xp, fp, xl, fl = [], [], [], [], [], []
    for v, T in enumerate((15, 12, 9)):
        lens = [T, T - 2, T]
        x = synth.reps(seed=900 + v, B=3, T=T)
        for b, n in enumerate(lens):
            x[b, :, n:] = 0
        flen = [2, 1, 2]
        f = synth.reps(seed=950 + v, B=3, T=2)
        f[1, :, 1:] = 0
        xs.append(x.to(DEV)); fs.append(f.to(DEV))
        xp.append(synth.padding_mask(lens).to(DEV)); fp.append(synth.padding_mask(flen).to(DEV))
        xl.append(lens); fl.append(flen)
    args = (xs if mod != "Flow" else None, fs if mod != "RGB" else None, xl, fl, 'Prototypes',
            xp if mod != "Flow" else None, fp if mod != "RGB" else None, None)
    old = tmod._TTA_MERGE
    try:
        with torch.no_grad():
            tmod._TTA_MERGE = True
            e1, a1 = m(*args)
            tmod._TTA_MERGE = False
            e0, a0 = m(*args)
    finally:
        tmod._TTA_MERGE = old
    assert a1.shape == a0.shape
    for v in range(3):
        assert maxabs(e1[v], e0[v].cpu().numpy()) <= 1e-4          # embeddings of magnitude ~3: split-K order differs with M
    assert maxabs(a1, a0.cpu().numpy()) <= 1e-5


@pytest.mark.parametrize("C", [2, 3])
def test_loss_logits_and_grads_vs_golden(gpu, golden, C):
    from sais_amd.loss import calcNCELoss, cosine_logits_and_probs
    g = golden("temporal")
    key = f"loss/C{C}/"
    m = make_full(C, "RGB-Flow").train()
    lens = [int(v) for v in g[key + "lens"]]
    B, T = len(lens), 32
    x = synth.reps(seed=500 + C, B=B, T=T)
    f = synth.reps(seed=600 + C, B=B, T=T)
    for b, n in enumerate(lens):
        x[b, :, n:] = 0
        f[b, :, n:] = 0
    x = x.to(DEV).requires_grad_(True)
    f = f.to(DEV)
    pad = synth.padding_mask(lens).to(DEV)
    protos = protos_dev(C)
    lab = synth.labels(seed=700 + C, B=B, nclasses=C)
    emb, attn = m(x, f, lens, lens, 'Prototypes', pad, pad, None)
    loss = calcNCELoss(0, emb, lab, [f"vid_{i}" for i in range(B)], protos, None)
    loss.backward()
    sim, probs = cosine_logits_and_probs(emb, protos)
    assert maxabs(sim, g[key + "sim"]) <= LOGIT_TOL, maxabs(sim, g[key + "sim"])
    assert maxabs(probs, g[key + "probs"]) <= LOGIT_TOL
    assert abs(loss.item() - float(g[key + "loss"])) <= LOGIT_TOL
    assert maxabs(attn, g[key + "attn"]) <= ATTN_TOL
    # Gradients and ReLU gates.  This batch has 264 rows x 2048 FFN units x 4 layers x 2 streams = 4.3 M ReLU gates; a gate
    # whose pre-activation lies within fp32 rounding (~1e-6 relative) of zero is open in one correct fp32 evaluation and
    # closed in another (a few per forward: the reference's own summation order vs any other), and ONE flipped gate in the
    # CLS row of a short clip moves that clip's gradient by 1-4 % (measured: clips without a flipped gate agree to 2e-5,
    # clips with one to 1e-3 .. 4e-2; tests/golden multidomain fixture note).  So: every clip / tensor within GATE_REL
    # (one or two flipped gates), and the TYPICAL clip / tensor (median) within 1e-3 — the bar that a wrong kernel fails.
    GATE_REL = 8e-2
    gx = g[key + "grad_x"]
    per_clip = [rel_l2(x.grad[b], gx[b]) for b in range(B)]
    assert max(per_clip) <= GATE_REL and float(np.median(per_clip)) <= 1e-3, per_clip
    for k in protos.keys():
        assert rel_l2(protos[k].grad, g[key + f"grad_proto{k}"]) <= GRAD_REL
    P = dict(m.named_parameters())
    errs = {}
    for k in g.files:
        if k.startswith(key + "grad/"):
            n = k[len(key + "grad/"):]
            errs[n] = rel_l2(P[n].grad, g[k])
        elif k.startswith(key + "grad8/"):
            n = k[len(key + "grad8/"):]
            errs[n] = rel_l2(P[n].grad[:8], g[k])
    bad = {n: r for n, r in errs.items() if r > GATE_REL}
    assert not bad, bad
    assert float(np.median(list(errs.values()))) <= 2e-2, sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    # The tight check: the fp64 oracle evaluated WITH THE GATES THIS FORWARD TOOK (oracle.imposed_gates) — what is left is
    # the arithmetic of the backward kernels alone.  Measured: 1.5e-5 .. 1.7e-5 on every clip with 13 of 4.3 M gates differing.
    from oracle import sais_oracle as O
    from parity import hip_temporal_gates, parity_log
    gates = hip_temporal_gates(m, x, f, pad, pad)
    sd = {k: v.double().clone().requires_grad_(True) for k, v in synth.temporal_state_dict(seed=1).items()}
    pr = {k: v.double().clone().requires_grad_(True) for k, v in synth.prototypes(2, C).items()}
    xr = x.detach().cpu().double().requires_grad_(True)
    with O.imposed_gates(gates) as ig:
        e64, _ = O.temporal_forward(sd, xr, f.detach().cpu().double(), pad.cpu(), pad.cpu(), "RGB-Flow")
        O.nce_loss(e64, lab, pr).backward()
    assert sum(ig.mismatches) <= 200, ig.mismatches                       # a handful of 4.3 M, not a different network
    tight = [rel_l2(x.grad[b], xr.grad[b].numpy()) for b in range(B)]
    parity_log(f"temporal C{C}/d loss / d x per clip vs fp64 oracle at the same ReLU gates", max(tight), 1e-3)
    parity_log(f"temporal C{C}/ReLU gates that differ from the fp64 oracle (of 4.3 M)", sum(ig.mismatches), 200)
    assert max(tight) <= 1e-3, tight
    tight_p = {n: rel_l2(P[n].grad, sd[n].grad.numpy()) for n in errs}
    # (weight gradients: the dW GEMMs of the temporal layers round their fp32 operands to bf16: 1-2.5e-3 measured)
    assert max(tight_p.values()) <= 5e-3, sorted(tight_p.items(), key=lambda kv: -kv[1])[:5]
    # parameters the reference never touches (clip encoder, MIL heads, unused positions, linear2) get zero grad
    assert float(P["transEncoderClip.layers.0.linear1.weight"].grad.abs().max()) == 0.0
    assert float(P["frame_pos_embeddings.40"].grad.abs().max()) == 0.0
    assert float(P["linear2.weight"].grad.abs().max()) == 0.0


@pytest.mark.parametrize("modal", ["RGB", "RGB-Flow"])
def test_importance_head_vs_golden(gpu, golden, modal):
    """Optional -il head: importances, importance loss and the gradients it sends into the encoder."""
    from sais_amd.loss import calcImportanceLoss, calcNCELoss
    from sais_amd.temporal import fullModel
    g = golden("importance")
    key = modal + "/"
    m = fullModel('reps', 2, 'in_vs_out', 384, 'ViT', modalities=modal, importance_loss=True)
    m.load_state_dict(synth.temporal_state_dict(seed=3, importance=True), strict=True)
    m.dropout_p = 0.0
    m = m.to(DEV).train()
    lens = [int(v) for v in g[key + "lens"]]
    B, T = len(lens), 9
    x, f = synth.reps(seed=810, B=B, T=T), synth.reps(seed=811, B=B, T=T)
    for b, n in enumerate(lens):
        x[b, :, n:] = 0
        f[b, :, n:] = 0
    x = x.to(DEV).requires_grad_(True)
    pad = synth.padding_mask(lens).to(DEV)
    protos = protos_dev(2)
    lab = torch.from_numpy(g[key + "labels"])
    target = torch.from_numpy(g[key + "target"]).to(DEV)
    imp, emb, attn = m(x, f.to(DEV), lens, lens, 'Prototypes', pad, pad, None)
    iloss = calcImportanceLoss(imp, target, pad, lab)
    loss = calcNCELoss(0, emb, lab, [f"v_{i}" for i in range(B)], protos, None) + iloss
    loss.backward()
    assert tuple(imp.shape) == (B, 1, T + 1, 1)
    assert maxabs(imp, g[key + "imp"]) <= 1e-4
    assert abs(iloss.item() - float(g[key + "iloss"])) <= 1e-5 and abs(loss.item() - float(g[key + "loss"])) <= LOGIT_TOL
    assert rel_l2(x.grad, g[key + "grad_x"]) <= GRAD_REL
    P = dict(m.named_parameters())
    bad = {k: rel_l2(P[k[len(key + "grad/"):]].grad, g[k]) for k in g.files if k.startswith(key + "grad/")}
    assert not {k: v for k, v in bad.items() if v > GRAD_REL}, bad
    nan = calcImportanceLoss(imp.detach(), target, pad, torch.ones(B, dtype=torch.long))
    assert torch.isnan(nan)                                   # no low-skill sample: mean of an empty tensor, as the reference


# ------------------------------------------------------------------ end-to-end composition
def test_e2e_config1_vs_golden(gpu, golden):
    """BASELINE config 1: B=1, T=16, 1-layer temporal encoder, RGB — logits within 1e-3 of the reference."""
    from sais_amd.loss import calcNCELoss, cosine_logits_and_probs
    g = golden("e2e")
    vit = make_vit()
    m = make_full(2, "RGB", nlayers=1)
    clips = synth.clips(seed=916, B=1, T=16).to(DEV)
    pad = synth.padding_mask([16]).to(DEV)
    protos = protos_dev(2)
    with torch.no_grad():
        reps = vit(clips.view(16, 3, 224, 224)).view(1, 1, 16, 384)
        emb, attn = m(reps, reps, [16], [16], 'Prototypes', pad, pad, None)
        sim, _ = cosine_logits_and_probs(emb, protos)
        loss = calcNCELoss(0, emb, synth.labels(seed=816, B=1), ["v_0"], protos, None)
    assert maxabs(reps, g["cfg1/reps"]) <= FEAT_REL * np.abs(g["cfg1/reps"]).max()
    assert maxabs(sim, g["cfg1/sim"]) <= LOGIT_TOL, maxabs(sim, g["cfg1/sim"])
    assert maxabs(attn, g["cfg1/attn"]) <= ATTN_TOL, maxabs(attn, g["cfg1/attn"])
    assert abs(loss.item() - float(g["cfg1/loss"])) <= LOGIT_TOL


def test_e2e_train_step_grads_vs_golden(gpu, golden):
    """Whole fwd+bwd graph: frames -> ViT -> temporal encoder -> prototype loss.
    Logits / loss / near-loss gradients are checked against the reference's golden vectors.  Gradients
    deep in the chain are ILL-CONDITIONED with respect to the features (measured with the oracle: a 1 %
    perturbation of the ViT features moves d frame_cls by 16 % through the ReLU gates), so the chain rule
    is checked stage by stage with the CPU oracle evaluated at the SAME intermediate tensors:
      temporal grads  vs  oracle temporal backward at the GPU's own ViT features,
      ViT grads       vs  oracle ViT backward driven by the GPU's own d loss / d features."""
    from oracle import sais_oracle as O
    from sais_amd.loss import calcNCELoss, cosine_logits_and_probs
    g = golden("e2e")
    vit = make_vit()                       # eval(), as the reference ran for the goldens: DropPath 0.1 is the identity there
    m = make_full(2, "RGB-Flow").train()
    B, T = 2, 4
    clips = synth.clips(seed=900 + T, B=B, T=T)
    fclips = synth.clips(seed=950 + T, B=B, T=T)
    frames = torch.cat([clips, fclips]).view(2 * B * T, 3, 224, 224)
    pad = synth.padding_mask([T] * B)
    protos = protos_dev(2)
    lab = synth.labels(seed=800 + T, B=B)
    reps = vit(frames.to(DEV))
    reps.retain_grad()
    r5 = reps.view(2, B, 1, T, 384)
    emb, attn = m(r5[0], r5[1], [T] * B, [T] * B, 'Prototypes', pad.to(DEV), pad.to(DEV), None)
    loss = calcNCELoss(0, emb, lab, ["a", "b"], protos, None)
    loss.backward()
    sim, _ = cosine_logits_and_probs(emb, protos)
    assert maxabs(sim, g["train/sim"]) <= LOGIT_TOL, maxabs(sim, g["train/sim"])
    assert abs(loss.item() - float(g["train/loss"])) <= LOGIT_TOL
    V, P = dict(vit.named_parameters()), dict(m.named_parameters())
    for k in protos.keys():                                   # near-loss gradients: well conditioned
        assert rel_l2(protos[k].grad, g[f"train/grad_proto{k}"]) <= GRAD_REL
    assert rel_l2(P["linear.bias"].grad, g["train/tgrad/linear.bias"]) <= GRAD_REL

    # stage 1: temporal backward at the GPU's own features
    reps_cpu = reps.detach().cpu().view(2, B, 1, T, 384)
    tsd = {k: v.clone().requires_grad_(True) for k, v in synth.temporal_state_dict(seed=1).items()}
    pr = {k: v.clone().requires_grad_(True) for k, v in synth.prototypes(2, 2).items()}
    rx, rf = reps_cpu[0].clone().requires_grad_(True), reps_cpu[1].clone().requires_grad_(True)
    e_ref, _ = O.temporal_forward(tsd, rx, rf, pad, pad, "RGB-Flow")
    O.nce_loss(e_ref, lab, pr).backward()
    bad = {}
    for n in ("linear.weight", "frame_cls", "frame_pos_embeddings.0", "frame_pos_embeddings.3",
              "transEncoderFrame.layers.0.self_attn.in_proj_weight", "transEncoderFrame.layers.3.norm2.bias",
              "transEncoderFrame.layers.1.linear1.weight", "transEncoderFrame.layers.2.linear2.bias"):
        r = rel_l2(P[n].grad, tsd[n].grad.numpy())
        if r > GRAD_REL:
            bad[n] = r
    dreps_ref = torch.cat([rx.grad, rf.grad]).reshape(2 * B * T, 384)
    r = rel_l2(reps.grad, dreps_ref.numpy())
    if r > GRAD_REL:
        bad["d loss / d reps"] = r
    assert not bad, bad

    # stage 2: ViT backward driven by the GPU's own upstream gradient
    vsd = {k: v.clone().requires_grad_(True) for k, v in synth.vit_state_dict(seed=0).items()}
    (O.vit_forward(vsd, frames) * reps.grad.detach().cpu()).sum().backward()
    bad = {}
    for n, q in V.items():
        r = rel_l2(q.grad, vsd[n].grad.numpy())
        if r > 6e-2:                      # 12 blocks of bf16 operands
            bad[n] = r
    assert not bad, bad


def test_sgd_step_matches_oracle_update(gpu):
    """One fused SGD step (lr 0.1, main.sh:27) moves every touched parameter by -lr*grad and refreshes the
    bf16 shadows: the next forward differs and equals a forward with manually updated weights."""
    from sais_amd.loss import calcNCELoss
    from sais_amd.optim import SGD
    m = make_full(2, "RGB").train()
    protos = protos_dev(2)
    lens = [6, 4]
    x = synth.reps(seed=41, B=2, T=6).to(DEV)
    pad = synth.padding_mask(lens).to(DEV)
    lab = synth.labels(seed=42, B=2)
    opt = SGD(list(m.parameters()) + list(protos.values()), lr=0.1, engines=[m])
    emb0, _ = m(x, None, lens, lens, 'Prototypes', pad, None, None)
    loss = calcNCELoss(0, emb0, lab, ["a", "b"], protos, None)
    opt.zero_grad()
    loss.backward()
    w_before = m.linear.weight.detach().clone()
    g_lin = m.linear.weight.grad.detach().clone()
    p_before, g_p = protos["0"].detach().clone(), protos["0"].grad.detach().clone()
    opt.step()
    assert torch.allclose(m.linear.weight, w_before - 0.1 * g_lin, atol=1e-6)
    assert torch.allclose(protos["0"], p_before - 0.1 * g_p, atol=1e-6)
    with torch.no_grad():
        emb1, _ = m(x, None, lens, lens, 'Prototypes', pad, None, None)
    assert (emb1 - emb0).abs().max().item() > 1e-4
    loss1 = calcNCELoss(0, emb1, lab, ["a", "b"], protos, None)
    assert loss1.item() < loss.item()


@pytest.mark.parametrize("rate,train", [(0.0, True), (0.2, True), (0.0, False)])
def test_block_level_c_calls_equal_the_per_launch_sequence(gpu, rate, train):
    """sais_vit_block_fwd / _bwd (one C call per Block: SURVEY 8b) issue the same launches in the same order as the
    per-launch Python sequence: features bit-identical, gradients equal up to the order of the fp32 atomics of the dW
    kernels; training (everything saved), DropPath, and inference (in place on the residual stream, GELU(u) in the
    workspace).  48 frames: the LayerNorm-fused regime the block entries dispatch at M >= 8192."""
    from sais_amd.vit import vit_small

    def run(block_calls):
        v = vit_small(patch_size=16, drop_path_rate=rate, depth=3)
        v.load_state_dict(synth.vit_state_dict(seed=0, depth=3), strict=True)
        v = v.to(DEV)
        v.block_calls = block_calls
        v.prune_last_block = False                    # all three blocks through the block entries
        v.drop_path_seed = 9
        x = synth.clips(seed=951, B=1, T=48)[0].to(DEV)
        if not train:
            with torch.no_grad():
                return v.eval()(x), None
        v.train()
        w = synth.reps(seed=952, B=1, T=48)[0, 0].to(DEV)
        feat = v(x)
        (feat * w).sum().backward()
        return feat.detach(), v.flat.grad.clone()

    f1, g1 = run(True)
    f0, g0 = run(False)
    assert torch.equal(f1, f0)
    if train:
        assert float(g0.abs().max()) > 0
        assert float((g1 - g0).norm() / g0.norm()) <= 1e-5


@pytest.mark.parametrize("frames,prune", [(128, True), (64, False)])
def test_grouped_weight_gradient_launches_equal_one_per_block(gpu, monkeypatch, frames, prune):
    """SAIS_DW_GROUP = G (LABNOTES R6.8): the weight / bias gradients of G blocks in ONE grouped launch (sais_vit_blocks_dw; the
    operands of a deferred block — its own du / d(mid) / dqkv, the bf16 gradient that entered it — stay alive until then) must equal
    one launch per block up to fp32 summation order (the number of M-splits changes with G), on both backward paths: the block-level
    C calls and the per-launch Python sequence.  depth 5, DropPath on: groups of 2 + 2 + 1 (or 2 + 2 with the pruned last block),
    of 4 + 1, of 5.  128 frames = 788 steps of 32 rows (the 192 x 384 kernel at every G), 64 frames = 394 (G = 1 falls back)."""
    from sais_amd.vit import vit_small

    def run(G, block_calls):
        monkeypatch.setenv("SAIS_DW_GROUP", str(G))
        v = vit_small(patch_size=16, drop_path_rate=0.2, depth=5)
        v.load_state_dict(synth.vit_state_dict(seed=0, depth=5), strict=True)
        v = v.to(DEV).train()
        v.block_calls = block_calls
        v.prune_last_block = prune
        v.drop_path_seed = 9
        x = synth.clips(seed=961, B=1, T=frames)[0].to(DEV)
        w = synth.reps(seed=962, B=1, T=frames)[0, 0].to(DEV)
        feat = v(x)
        (feat * w).sum().backward()
        return feat.detach(), v.flat.grad.clone()

    f_ref, g_ref = run(1, True)
    assert float(g_ref.abs().max()) > 0
    for G, bc in ((2, True), (4, True), (10, True), (2, False), (5, False)):
        f, g = run(G, bc)
        assert torch.equal(f, f_ref)
        assert float((g - g_ref).norm() / g_ref.norm()) <= 1e-5, (G, bc)


@pytest.mark.parametrize("frames,rate", [(6, 0.0), (48, 0.2)])
def test_pruned_last_block_equals_full_compute(gpu, frames, rate):
    """forward() returns norm(x)[:, 0]: the last block runs on the CLS rows / the CLS query only (sais_amd.vit,
    prune_last_block).  Features and EVERY parameter gradient must equal the run that computes all rows (both dispatch
    regimes: stand-alone kernels at 6 frames, row-owning / fused kernels at 48; with DropPath draws shared)."""
    from sais_amd.vit import vit_small

    def run(prune):
        v = vit_small(patch_size=16, drop_path_rate=rate, depth=3)
        v.load_state_dict(synth.vit_state_dict(seed=0, depth=3), strict=True)
        v = v.to(DEV).train()
        v.prune_last_block = prune
        v.drop_path_seed = 3
        x = synth.clips(seed=941, B=1, T=frames)[0].to(DEV)
        w = synth.reps(seed=942, B=1, T=frames)[0, 0].to(DEV)
        feat = v(x)
        (feat * w).sum().backward()
        with torch.no_grad():
            ev = v.eval()(x)
        return feat.detach(), {n: q.grad.detach().clone() for n, q in v.named_parameters()}, ev

    f1, g1, e1 = run(True)
    f0, g0, e0 = run(False)
    scale = float(f0.abs().max())
    assert float((f1 - f0).abs().max()) <= 5e-3 * scale and float((e1 - e0).abs().max()) <= 5e-3 * scale
    worst = {n: float((g1[n] - g0[n]).norm() / g0[n].norm().clamp_min(1e-12)) for n in g0}
    bad = {n: r for n, r in worst.items() if r > 1e-2}
    assert not bad, bad


def test_outlier_weights_logits_within_the_bar(gpu, golden):
    """Weights with the dynamic range of a TRAINED DINO checkpoint (synth.vit_state_dict_outlier: massive-activation
    residual channels up to |x| ~ 70, LayerNorm gains x20-x50, sharper attention) instead of an initialiser's: the
    class logits must still match the reference within the 1e-3 north-star bar."""
    from sais_amd.loss import cosine_logits_and_probs
    from sais_amd.vit import vit_small
    g = golden("outlier")
    vit = vit_small(patch_size=16, drop_path_rate=0.0)
    vit.load_state_dict(synth.vit_state_dict_outlier(seed=3), strict=True)
    vit = vit.to(DEV).eval()
    m = make_full(2, "RGB")
    B, T = 2, 8
    clips = synth.clips(seed=977, B=B, T=T).to(DEV)
    lens = [T, T - 3]
    pad = synth.padding_mask(lens).to(DEV)
    with torch.no_grad():
        reps = vit(clips.view(B * T, 3, 224, 224)).view(B, 1, T, 384)
        emb, attn = m(reps, None, lens, None, 'Prototypes', pad, None, None)
        sim, _ = cosine_logits_and_probs(emb, protos_dev(2))
    dfeat = maxabs(reps, g["reps"])
    assert dfeat <= FEAT_REL * np.abs(g["reps"]).max(), dfeat
    assert maxabs(sim, g["sim"]) <= LOGIT_TOL, maxabs(sim, g["sim"])
    assert maxabs(attn, g["attn"]) <= ATTN_TOL, maxabs(attn, g["attn"])
    # the same weights at the row-kernel dispatch (M >= 8192: 48 frames): logits agree with the small-batch path
    big = synth.clips(seed=977, B=6, T=8)
    big[:2] = clips.cpu()
    with torch.no_grad():
        reps_big = vit(big.view(48, 3, 224, 224).to(DEV))[:16].view(B, 1, T, 384)
        emb2, _ = m(reps_big, None, lens, None, 'Prototypes', pad, None, None)
        sim2, _ = cosine_logits_and_probs(emb2, protos_dev(2))
    assert maxabs(sim2, g["sim"]) <= LOGIT_TOL, maxabs(sim2, g["sim"])


@pytest.mark.parametrize("modal", ["RGB", "RGB-Flow"])
def test_multiple_snippets_per_clip_vs_golden(gpu, golden, modal):
    """nsnippets = 3 (prepare_model.py:179-221: every (clip, snippet) is one encoder sequence; :381-382: mean of the
    ReLU'd CLS rows over the snippets): embeddings, the [B*ns, S, S] attention map, loss and gradients vs the reference."""
    import make_golden as MG
    from sais_amd.loss import calcNCELoss
    g = golden("snippets")
    x, f, pad, lab = MG.snippet_inputs()
    m = make_full(2, modal).train()
    protos = protos_dev(2)
    xd = x.to(DEV).requires_grad_(True)
    fd = f.to(DEV).requires_grad_(True)
    emb, attn = m(xd, fd, None, None, 'Prototypes', pad.to(DEV), pad.to(DEV), None)
    loss = calcNCELoss(0, emb, lab, ["a", "b"], protos, None)
    loss.backward()
    key = modal + "/"
    assert tuple(attn.shape) == (6, 7, 7)
    assert maxabs(emb, g[key + "emb"]) <= 2e-3 * max(1.0, np.abs(g[key + "emb"]).max())
    assert maxabs(attn, g[key + "attn"]) <= ATTN_TOL
    assert abs(loss.item() - float(g[key + "loss"])) <= LOGIT_TOL
    assert rel_l2(xd.grad, g[key + "grad_x"]) <= GRAD_REL and tuple(xd.grad.shape) == (2, 3, 6, 384)
    if modal == "RGB-Flow":
        assert rel_l2(fd.grad, g[key + "grad_f"]) <= GRAD_REL
    P = dict(m.named_parameters())
    for k in g.files:
        if k.startswith(key + "grad/"):
            assert rel_l2(P[k[len(key + "grad/"):]].grad, g[k]) <= GRAD_REL, k


def test_multidomain_head_selects_linear_or_linearb_per_sample(gpu, golden):
    """fullModel(domain 'NH_02+HMH_01', 'RGB-Flow') (prepare_model.py:405-414): samples whose domain is not 'NH_02' go
    through linearB — embeddings, loss and both heads' gradients vs the reference's own two-domain run; TTA list form too."""
    import make_golden as MG
    from sais_amd.loss import calcNCELoss
    from sais_amd.temporal import fullModel
    g = golden("multidomain")
    x, f, pad, lab, domains, lens = MG.multidomain_inputs()
    m = fullModel('reps', 2, 'NH_02+HMH_01', 384, 'ViT', modalities='RGB-Flow')
    m.load_state_dict(synth.temporal_state_dict(seed=1, multidomain=True), strict=True)
    m.dropout_p = 0.0
    m = m.to(DEV).train()
    protos = protos_dev(2)
    xd, fd = x.to(DEV).requires_grad_(True), f.to(DEV).requires_grad_(True)
    emb, attn = m(xd, fd, lens, lens, 'Prototypes', pad.to(DEV), pad.to(DEV), domains)
    loss = calcNCELoss(0, emb, lab, list("abcd"), protos, domains)
    loss.backward()
    assert maxabs(emb, g["emb"]) <= 2e-3 * max(1.0, np.abs(g["emb"]).max())
    assert maxabs(attn, g["attn"]) <= ATTN_TOL and abs(loss.item() - float(g["loss"])) <= LOGIT_TOL
    assert rel_l2(xd.grad, g["grad_x"]) <= GRAD_REL and rel_l2(fd.grad, g["grad_f"]) <= GRAD_REL
    P = dict(m.named_parameters())
    for k in g.files:
        if k.startswith("grad/"):
            assert rel_l2(P[k[5:]].grad, g[k]) <= GRAD_REL, (k, rel_l2(P[k[5:]].grad, g[k]))
    xs, fs = [x[:, :, :7], x[:, :, :5], x[:, :, :3]], [f[:, :, :7], f[:, :, :5], f[:, :, :3]]
    pads = [synth.padding_mask([min(l, n) for l in lens])[:, :, :n + 1].to(DEV) for n in (7, 5, 3)]
    with torch.no_grad():
        embs, _ = m.eval()([t.contiguous().to(DEV) for t in xs], [t.contiguous().to(DEV) for t in fs], None, None,
                           'Prototypes', pads, pads, domains)
    for v in range(3):
        assert maxabs(embs[v], g[f"tta/emb{v}"]) <= 2e-3 * max(1.0, np.abs(g[f"tta/emb{v}"]).max())
    with pytest.raises(ValueError):
        m(xd, fd, lens, lens, 'Prototypes', pad.to(DEV), pad.to(DEV), None)        # a multi-domain model needs `domains`


def test_mil_forward_vs_golden(gpu, golden):
    """task 'MIL', inference (prepare_model.py:356-361,452-488,131-148): clip-level encoder over the snippet representations
    and the gated-attention MIL head against the reference's own outputs."""
    import make_golden as MG
    from sais_amd import _lib as L
    g = golden("mil")
    x, f, pad, _ = MG.snippet_inputs()
    m = make_full(2, "RGB-Flow")
    seq, reps, logits, att = m(x.to(DEV), f.to(DEV), None, None, 'MIL', pad.to(DEV), pad.to(DEV), None)
    assert tuple(seq.shape) == (3, 2, 384) and tuple(reps.shape) == (2, 3, 384) and tuple(logits.shape) == (2, 2)
    assert maxabs(seq, g["snip_sequence"]) <= 1e-3
    assert maxabs(reps, g["snip_reps"]) <= 2e-3 * max(1.0, np.abs(g["snip_reps"]).max())
    assert maxabs(logits, g["logits"]) <= LOGIT_TOL, maxabs(logits, g["logits"])
    for c in range(2):
        assert maxabs(att[c], g[f"attention{c}"]) <= 1e-3
    with pytest.raises(NotImplementedError):
        m.train()(x.to(DEV), f.to(DEV), None, None, 'MIL', pad.to(DEV), pad.to(DEV), None)
    with pytest.raises(NotImplementedError):
        make_full(2, "RGB")(x.to(DEV), None, None, None, 'MIL', pad.to(DEV), None, None)
    assert L is not None
