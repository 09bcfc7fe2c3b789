"""Parity at the sizes the benchmark runs (VERDICT r1 'weak' 1-2): the kernels `sais_gemm_nt` / `sais_gemm_tn_grouped`
dispatch for M >= 8192 (persistent eight-wave NT kernel, wide dW kernel, LN-fused row-owning GEMMs) only run at
F >= 42 frames, and the hipGraph replay of the training step is what `bench.py` times.

  * stage-by-stage oracle check of one training step at F = 64 (M = 12 608) and F = 256 (config 2, M = 50 432):
    ViT features / logits / loss vs the CPU oracle, temporal gradients at the GPU's own features, ViT gradients
    driven by the GPU's own d loss / d features (the chain is ill-conditioned end to end: test_model_gpu.py);
  * config-2 step: eager launch vs GraphedStep replay from identical weights -> same loss, same gradients, same
    updated weights.

Tolerances: logits 1e-3 max-abs (north star), features 2e-2 of max|ref|, gradients 4e-2 / 6e-2 relative L2 (bf16 MFMA
operands, fp32 accumulation), graph-vs-eager gradients 1e-5 relative L2 (fp32 atomics reorder the dW sums).
"""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
# bars at the benchmark's own size, tightened in round 3 to ~3x the worst values observed there (profiles/r03_a_parity_worst.json:
# features 1.0 %, ViT parameter gradients 0.7 %, temporal gradients 0.1 %, d loss / d features 0.13 %)
LOGIT_TOL, FEAT_REL, GRAD_REL, VIT_GRAD_REL = 1e-3, 2e-2, 1e-2, 2e-2


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return True


def rel_l2(a, b):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-12))


def _models():
    from sais_amd.temporal import fullModel
    from sais_amd.vit import vit_small
    vit = vit_small(patch_size=16, drop_path_rate=0.0)
    vit.load_state_dict(synth.vit_state_dict(seed=0), strict=True)
    m = fullModel('reps', 2, 'in_vs_out', 384, 'ViT', modalities='RGB')
    m.load_state_dict(synth.temporal_state_dict(seed=1), strict=True)
    m.dropout_p = 0.0
    protos = torch.nn.ParameterDict({k: torch.nn.Parameter(v.clone().to(DEV)) for k, v in synth.prototypes(2, 2).items()})
    return vit.to(DEV).train(), m.to(DEV).train(), protos


@pytest.mark.parametrize("B,T", [(2, 32), (8, 32)])
def test_train_step_stagewise_vs_oracle_at_benchmark_dispatch(gpu, B, T):
    from oracle import sais_oracle as O
    from sais_amd.loss import calcNCELoss, cosine_logits_and_probs
    vit, m, protos = _models()
    F = B * T
    assert F * 197 >= 8192                                    # the big-M kernels are the ones under test
    clips = synth.clips(seed=1000 + F, B=B, T=T)
    frames = clips.view(F, 3, 224, 224)
    lens = [T - (3 * b) % 7 for b in range(B)]                # ragged clips: key-padding mask in play
    lens[0] = T
    pad = synth.padding_mask(lens)
    lab = synth.labels(seed=1100 + F, B=B)
    reps = vit(frames.to(DEV))
    reps.retain_grad()
    emb, attn = m(reps.view(B, 1, T, 384), None, lens, None, 'Prototypes', pad.to(DEV), None, None)
    loss = calcNCELoss(0, emb, lab, [f"v{b}" for b in range(B)], protos, None)
    loss.backward()
    sim, _ = cosine_logits_and_probs(emb, protos)
    torch.cuda.synchronize()

    # forward: features, logits, attention, loss vs the oracle on the same frames
    vsd = {k: v.clone().requires_grad_(True) for k, v in synth.vit_state_dict(seed=0).items()}
    tsd = {k: v.clone().requires_grad_(True) for k, v in synth.temporal_state_dict(seed=1).items()}
    pr = {k: v.clone().requires_grad_(True) for k, v in synth.prototypes(2, 2).items()}
    # the oracle ViT runs in chunks of 32 frames (bounded host memory); its backward is driven chunk by chunk by the GPU's
    # own d loss / d features, so the parameter gradients accumulate over all frames (stage 2 below)
    dreps_gpu = reps.grad.detach().cpu()
    parts = []
    for i in range(0, F, 32):
        r = O.vit_forward(vsd, frames[i:i + 32])
        parts.append(r.detach())
        (r * dreps_gpu[i:i + 32]).sum().backward()
    reps_ref = torch.cat(parts)
    with torch.no_grad():
        e_ref, a_ref = O.temporal_forward(tsd, reps_ref.detach().view(B, 1, T, 384), None, pad, None, "RGB")
        sim_ref = O.cosine_logits(e_ref, pr)
        loss_ref = O.nce_loss(e_ref, lab, pr)
    from parity import parity_log
    tag = f"step[B{B},T{T}]/"
    dfeat = (reps.detach().cpu() - reps_ref.detach()).abs().max().item()
    parity_log(tag + "features max-abs / max|ref|", dfeat / reps_ref.detach().abs().max().item(), FEAT_REL)
    assert dfeat <= FEAT_REL * reps_ref.detach().abs().max().item(), dfeat
    dlogit = (sim.cpu() - sim_ref).abs().max().item()
    parity_log(tag + "cosine logits max-abs", dlogit, LOGIT_TOL)
    assert dlogit <= LOGIT_TOL, dlogit
    dattn = (attn.cpu() - a_ref).abs().max().item()
    parity_log(tag + "attention map max-abs", dattn, 2e-3)
    assert dattn <= 2e-3
    parity_log(tag + "loss abs", abs(loss.item() - loss_ref.item()), LOGIT_TOL)
    assert abs(loss.item() - loss_ref.item()) <= LOGIT_TOL

    # stage 1: temporal backward at the GPU's own features
    P = dict(m.named_parameters())
    rx = reps.detach().cpu().view(B, 1, T, 384).clone().requires_grad_(True)
    e1, _ = O.temporal_forward(tsd, rx, None, pad, None, "RGB")
    O.nce_loss(e1, lab, pr).backward()
    bad = {}
    for n in ("linear.weight", "linear.bias", "frame_cls", "frame_pos_embeddings.0", f"frame_pos_embeddings.{T - 1}",
              "transEncoderFrame.layers.0.self_attn.in_proj_weight", "transEncoderFrame.layers.3.norm2.bias",
              "transEncoderFrame.layers.1.linear1.weight", "transEncoderFrame.layers.2.linear2.bias"):
        r = rel_l2(P[n].grad, tsd[n].grad)
        parity_log(tag + "temporal parameter gradients, worst tensor rel-L2", r, GRAD_REL)
        if r > GRAD_REL:
            bad[n] = r
    for k in protos.keys():
        r = rel_l2(protos[k].grad, pr[k].grad)
        if r > GRAD_REL:
            bad["proto" + k] = r
    r = rel_l2(reps.grad, rx.grad.reshape(F, 384))
    parity_log(tag + "d loss / d features rel-L2", r, GRAD_REL)
    if r > GRAD_REL:
        bad["d loss / d reps"] = r
    assert not bad, bad
    # ... and tightly: the fp64 oracle at the ReLU gates the HIP forward took (tests/test_model_gpu.py explains why)
    from parity import hip_temporal_gates
    gates = hip_temporal_gates(m, reps.detach().view(B, 1, T, 384), None, pad.to(DEV), None)
    tsd64 = {k: v.double().clone().requires_grad_(True) for k, v in synth.temporal_state_dict(seed=1).items()}
    pr64 = {k: v.double().clone() for k, v in synth.prototypes(2, 2).items()}
    rx64 = reps.detach().cpu().double().view(B, 1, T, 384).clone().requires_grad_(True)
    with O.imposed_gates(gates) as ig:
        e64, _ = O.temporal_forward(tsd64, rx64, None, pad, None, "RGB")
        O.nce_loss(e64, lab, pr64).backward()
    tight = max(rel_l2(reps.grad.view(B, T, 384)[b], rx64.grad[b, 0].numpy()) for b in range(B))
    parity_log(tag + "d loss / d features per clip vs fp64 oracle at the same ReLU gates", tight, 1e-3)
    parity_log(tag + "ReLU gates that differ from the fp64 oracle", sum(ig.mismatches), 200)
    assert tight <= 1e-3 and sum(ig.mismatches) <= 200, (tight, ig.mismatches)

    # stage 2: ViT backward driven by the GPU's own upstream gradient (all frames: parameter gradients sum over them)
    bad = {}
    worst = 0.0
    for n, q in vit.named_parameters():
        r = rel_l2(q.grad, vsd[n].grad)
        worst = max(worst, r)
        if r > VIT_GRAD_REL:
            bad[n] = r
    parity_log(tag + "ViT parameter gradients, worst tensor rel-L2 (150 tensors)", worst, VIT_GRAD_REL)
    assert not bad, bad


def test_two_stream_train_step_stagewise_vs_oracle_at_config4_size(gpu):
    """BASELINE config 4 at its bench size: 8 clips x 32 frames x 2 streams (RGB + flow frames), 512 frames through the ViT
    in ONE pass (M = 100 864 rows: two whole rounds of the eight-wave row tiles), the temporal encoder once per stream,
    streams fused by add (prepare_model.py:412), loss, backward — checked stage by stage like the config-2 test."""
    from oracle import sais_oracle as O
    from parity import parity_log
    from sais_amd.loss import calcNCELoss, cosine_logits_and_probs
    from sais_amd.temporal import fullModel
    B, T = 8, 32
    vit, _, protos = _models()
    m = fullModel('reps', 2, 'in_vs_out', 384, 'ViT', modalities='RGB-Flow')
    m.load_state_dict(synth.temporal_state_dict(seed=1), strict=True)
    m.dropout_p = 0.0
    m = m.to(DEV).train()
    F = 2 * B * T
    frames = torch.cat([synth.clips(seed=2000, B=B, T=T), synth.clips(seed=2001, B=B, T=T)]).view(F, 3, 224, 224)
    lens = [T - (5 * b) % 9 for b in range(B)]
    lens[0] = T
    pad = synth.padding_mask(lens)
    lab = synth.labels(seed=2100, B=B)
    reps = vit(frames.to(DEV))
    reps.retain_grad()
    r5 = reps.view(2, B, 1, T, 384)
    emb, attn = m(r5[0], r5[1], lens, lens, 'Prototypes', pad.to(DEV), pad.to(DEV), None)
    loss = calcNCELoss(0, emb, lab, [f"v{b}" for b in range(B)], protos, None)
    loss.backward()
    sim, _ = cosine_logits_and_probs(emb, protos)
    torch.cuda.synchronize()
    vsd = {k: v.clone().requires_grad_(True) for k, v in synth.vit_state_dict(seed=0).items()}
    tsd = {k: v.clone().requires_grad_(True) for k, v in synth.temporal_state_dict(seed=1).items()}
    pr = {k: v.clone().requires_grad_(True) for k, v in synth.prototypes(2, 2).items()}
    dreps_gpu = reps.grad.detach().cpu()
    parts = []
    for i in range(0, F, 32):
        r = O.vit_forward(vsd, frames[i:i + 32])
        parts.append(r.detach())
        (r * dreps_gpu[i:i + 32]).sum().backward()
    reps_ref = torch.cat(parts)
    rr = reps_ref.view(2, B, 1, T, 384)
    with torch.no_grad():
        e_ref, a_ref = O.temporal_forward(tsd, rr[0], rr[1], pad, pad, "RGB-Flow")
        sim_ref, loss_ref = O.cosine_logits(e_ref, pr), O.nce_loss(e_ref, lab, pr)
    tag = "two-stream step[B8,T32]/"
    dfeat = (reps.detach().cpu() - reps_ref).abs().max().item() / reps_ref.abs().max().item()
    dlogit = (sim.cpu() - sim_ref).abs().max().item()
    dattn = (attn.cpu() - a_ref).abs().max().item()
    parity_log(tag + "features max-abs / max|ref|", dfeat, FEAT_REL)
    parity_log(tag + "cosine logits max-abs", dlogit, LOGIT_TOL)
    parity_log(tag + "attention map max-abs", dattn, 2e-3)
    assert dfeat <= FEAT_REL and dlogit <= LOGIT_TOL and dattn <= 2e-3, (dfeat, dlogit, dattn)
    assert abs(loss.item() - loss_ref.item()) <= LOGIT_TOL
    # stage 1: temporal backward (both streams) at the GPU's own features
    gx = reps.detach().cpu().view(2, B, 1, T, 384)
    rx, rf = gx[0].clone().requires_grad_(True), gx[1].clone().requires_grad_(True)
    e1, _ = O.temporal_forward(tsd, rx, rf, pad, pad, "RGB-Flow")
    O.nce_loss(e1, lab, pr).backward()
    r = rel_l2(reps.grad, torch.cat([rx.grad.reshape(B * T, 384), rf.grad.reshape(B * T, 384)]))
    parity_log(tag + "d loss / d features rel-L2", r, GRAD_REL)
    assert r <= GRAD_REL, r
    P = dict(m.named_parameters())
    worst = max(rel_l2(P[n].grad, tsd[n].grad) for n in (
        "linear.weight", "frame_cls", "frame_pos_embeddings.0", "transEncoderFrame.layers.0.self_attn.in_proj_weight",
        "transEncoderFrame.layers.3.norm2.bias", "transEncoderFrame.layers.1.linear1.weight"))
    # vs the fp32 oracle a ReLU gate within rounding of zero may be open in one evaluation and closed in the other, which moves
    # that clip's gradient by per cents: loose bar here, tight bar against the fp64 oracle at the HIP forward's
    # own gates below
    parity_log(tag + "temporal parameter gradients, worst tensor rel-L2", worst, 4e-2)
    assert worst <= 4e-2, worst
    from parity import hip_temporal_gates
    gd = reps.detach().view(2, B, 1, T, 384)
    gates = hip_temporal_gates(m, gd[0], gd[1], pad.to(DEV), pad.to(DEV))
    tsd64 = {k: v.double().clone().requires_grad_(True) for k, v in synth.temporal_state_dict(seed=1).items()}
    pr64 = {k: v.double().clone() for k, v in synth.prototypes(2, 2).items()}
    rx64, rf64 = gx[0].double().clone().requires_grad_(True), gx[1].double().clone().requires_grad_(True)
    with O.imposed_gates(gates) as ig:
        e64, _ = O.temporal_forward(tsd64, rx64, rf64, pad, pad, "RGB-Flow")
        O.nce_loss(e64, lab, pr64).backward()
    tight = max(rel_l2(P[n].grad, tsd64[n].grad) for n in (
        "linear.weight", "frame_cls", "frame_pos_embeddings.0", "transEncoderFrame.layers.0.self_attn.in_proj_weight",
        "transEncoderFrame.layers.3.norm2.bias", "transEncoderFrame.layers.1.linear1.weight"))
    parity_log(tag + "temporal parameter gradients vs fp64 oracle at the same ReLU gates, worst tensor", tight, 5e-3)
    parity_log(tag + "ReLU gates that differ from the fp64 oracle", sum(ig.mismatches), 400)
    assert tight <= 5e-3 and sum(ig.mismatches) <= 400, (tight, ig.mismatches)
    # stage 2: ViT backward over all 512 frames
    worst = max(rel_l2(q.grad, vsd[n].grad) for n, q in vit.named_parameters())
    parity_log(tag + "ViT parameter gradients, worst tensor rel-L2 (150 tensors)", worst, VIT_GRAD_REL)
    assert worst <= VIT_GRAD_REL, worst


def test_benchmarked_configuration_with_dropout_and_droppath_vs_oracle(gpu):
    """the configuration bench.py TIMES — B = 8, T = 32 (M = 50 432), train() with temporal dropout 0.1
    AND ViT DropPath 0.1 — against the oracle fed the SAME draws: the per-sample DropPath factors the forward used
    (`last_droppath_scales`) and the dropout masks regenerated from the forward's RNG state (`dropout_masks`).  Forward:
    features, cosine logits, attention map, loss.  Backward, stage-wise as above: temporal gradients at the GPU's own
    features (same masks), ViT parameter gradients driven by the GPU's own d loss / d features (same DropPath factors)."""
    from oracle import sais_oracle as O
    from parity import parity_log
    from sais_amd.loss import calcNCELoss, cosine_logits_and_probs
    from sais_amd.temporal import fullModel
    from sais_amd.vit import vit_small
    B, T, C = 8, 32, 2
    F, S = B * T, T + 1
    vit = vit_small(patch_size=16, drop_path_rate=0.1)
    vit.load_state_dict(synth.vit_state_dict(seed=0), strict=True)
    vit = vit.to(DEV).train()
    vit.drop_path_seed = 7919
    m = fullModel('reps', C, 'in_vs_out', 384, 'ViT', modalities='RGB')
    m.load_state_dict(synth.temporal_state_dict(seed=1), strict=True)
    m = m.to(DEV).train()
    assert m.dropout_p == 0.1                                # the reference's nn.TransformerEncoderLayer default
    m.dropout_seed = 5
    protos = torch.nn.ParameterDict({k: torch.nn.Parameter(v.clone().to(DEV)) for k, v in synth.prototypes(2, C).items()})
    frames = synth.clips(seed=3000, B=B, T=T).view(F, 3, 224, 224)
    lens = [T] * B                                            # the benchmark's clips are full length
    pad = synth.padding_mask(lens)
    lab = synth.labels(seed=3001, B=B, nclasses=C)
    reps = vit(frames.to(DEV))
    reps.retain_grad()
    emb, attn = m(reps.view(B, 1, T, 384), None, lens, None, 'Prototypes', pad.to(DEV), None, None)
    loss = calcNCELoss(0, emb, lab, [f"v{b}" for b in range(B)], protos, None)
    loss.backward()
    sim, _ = cosine_logits_and_probs(emb, protos)
    torch.cuda.synchronize()
    # the draws this step used
    sc = vit.last_droppath_scales.view(24, F, 197)
    assert bool((sc == sc[:, :, :1]).all())
    fac = sc[:, :, 0].cpu()                                   # [24, F] per-sample branch factors
    assert float((fac == 0).sum()) >= 1                       # something was dropped
    st = m.last_dropout_state
    drop = {"rgb": [{k: v.cpu() for k, v in lm.items()} for lm in m.dropout_masks(st, B, S, stream=0)]}

    vsd = {k: v.clone().requires_grad_(True) for k, v in synth.vit_state_dict(seed=0).items()}
    tsd = {k: v.clone().requires_grad_(True) for k, v in synth.temporal_state_dict(seed=1).items()}
    pr = {k: v.clone().requires_grad_(True) for k, v in synth.prototypes(2, C).items()}
    dreps_gpu = reps.grad.detach().cpu()
    parts = []
    for i in range(0, F, 32):
        r = O.vit_forward(vsd, frames[i:i + 32], droppath=fac[:, i:i + 32])
        parts.append(r.detach())
        (r * dreps_gpu[i:i + 32]).sum().backward()
    reps_ref = torch.cat(parts)
    with torch.no_grad():
        e_ref, a_ref = O.temporal_forward(tsd, reps_ref.view(B, 1, T, 384), None, pad, None, "RGB", drop=drop, p=m.dropout_p)
        sim_ref, loss_ref = O.cosine_logits(e_ref, pr), O.nce_loss(e_ref, lab, pr)
    tag = "benchmarked step[B8,T32,dropout 0.1,DropPath 0.1]/"
    dfeat = (reps.detach().cpu() - reps_ref).abs().max().item() / reps_ref.abs().max().item()
    dlogit = (sim.cpu() - sim_ref).abs().max().item()
    dattn = (attn.cpu() - a_ref).abs().max().item()
    dloss = abs(loss.item() - loss_ref.item())
    parity_log(tag + "features max-abs / max|ref|", dfeat, FEAT_REL)
    parity_log(tag + "cosine logits max-abs", dlogit, LOGIT_TOL)
    parity_log(tag + "attention map max-abs (the dropped map)", dattn, 2e-3)
    parity_log(tag + "loss abs", dloss, LOGIT_TOL)
    assert dfeat <= FEAT_REL and dlogit <= LOGIT_TOL and dattn <= 2e-3 and dloss <= LOGIT_TOL, (dfeat, dlogit, dattn, dloss)
    # stage 1: temporal backward at the GPU's own features, same masks
    rx = reps.detach().cpu().view(B, 1, T, 384).clone().requires_grad_(True)
    e1, _ = O.temporal_forward(tsd, rx, None, pad, None, "RGB", drop=drop, p=m.dropout_p)
    O.nce_loss(e1, lab, pr).backward()
    r = rel_l2(reps.grad, rx.grad.reshape(F, 384))
    parity_log(tag + "d loss / d features rel-L2", r, 4e-2)
    assert r <= 4e-2, r                                       # a flipped ReLU gate moves one clip's gradient
    P = dict(m.named_parameters())
    grads = {n: rel_l2(P[n].grad, tsd[n].grad) for n in (
        "linear.weight", "frame_cls", "frame_pos_embeddings.0", f"frame_pos_embeddings.{T - 1}",
        "transEncoderFrame.layers.0.self_attn.in_proj_weight", "transEncoderFrame.layers.3.norm2.bias",
        "transEncoderFrame.layers.1.linear1.weight", "transEncoderFrame.layers.2.linear2.bias")}
    parity_log(tag + "temporal parameter gradients, worst tensor rel-L2", max(grads.values()), 4e-2)
    assert max(grads.values()) <= 4e-2, grads
    # stage 2: ViT backward with the same DropPath factors
    worst = {n: rel_l2(q.grad, vsd[n].grad) for n, q in vit.named_parameters()}
    parity_log(tag + "ViT parameter gradients, worst tensor rel-L2 (150 tensors)", max(worst.values()), VIT_GRAD_REL)
    assert max(worst.values()) <= VIT_GRAD_REL, {k: v for k, v in worst.items() if v > VIT_GRAD_REL}


def test_outlier_weights_at_the_eight_wave_dispatch(gpu):
    """The DINO-like outlier weights (synth.vit_state_dict_outlier: residual |x| ~ 37, LayerNorm outputs ~ 150, attention
    logits up to +-30) at M = 146 x 197 = 28 762 rows — the eight-wave row tile and the persistent 128 x 128 kernels the
    benchmark dispatches — against the oracle on the same frames; the deviation is logged."""
    from oracle import sais_oracle as O
    from parity import parity_log
    from sais_amd.loss import cosine_logits_and_probs
    from sais_amd.temporal import fullModel
    from sais_amd.vit import vit_small
    F, B, T = 146, 4, 32                                      # 146 frames through the ViT, the first 128 form 4 clips
    vit = vit_small(patch_size=16, drop_path_rate=0.0)
    sdo = synth.vit_state_dict_outlier(seed=3)
    vit.load_state_dict(sdo, strict=True)
    vit = vit.to(DEV).eval()
    m = fullModel('reps', 2, 'in_vs_out', 384, 'ViT', modalities='RGB')
    m.load_state_dict(synth.temporal_state_dict(seed=1), strict=True)
    m = m.to(DEV).eval()
    protos = {k: v.to(DEV) for k, v in synth.prototypes(2, 2).items()}
    frames = synth.clips(seed=3100, B=1, T=F)[0]
    lens = [T, T - 5, T, T - 11]
    pad = synth.padding_mask(lens)
    with torch.no_grad():
        reps = vit(frames.to(DEV))
        emb, attn = m(reps[:B * T].view(B, 1, T, 384), None, lens, None, 'Prototypes', pad.to(DEV), None, None)
        sim, _ = cosine_logits_and_probs(emb, protos)
        ref = torch.cat([O.vit_forward(sdo, frames[i:i + 32]) for i in range(0, B * T, 32)])
        e_ref, a_ref = O.temporal_forward(synth.temporal_state_dict(seed=1), ref.view(B, 1, T, 384), None, pad, None, "RGB")
        sim_ref = O.cosine_logits(e_ref, synth.prototypes(2, 2))
    tag = "outlier weights[M 28762]/"
    dfeat = (reps[:B * T].cpu() - ref).abs().max().item() / ref.abs().max().item()
    dlogit = (sim.cpu() - sim_ref).abs().max().item()
    dattn = (attn.cpu() - a_ref).abs().max().item()
    parity_log(tag + "features max-abs / max|ref|", dfeat, 2e-2)
    parity_log(tag + "cosine logits max-abs", dlogit, LOGIT_TOL)
    parity_log(tag + "attention map max-abs", dattn, 2e-3)
    assert dfeat <= 2e-2 and dlogit <= LOGIT_TOL and dattn <= 2e-3, (dfeat, dlogit, dattn)


def test_config2_graph_replay_equals_eager_step(gpu):
    """The code path bench.py times: one config-2 step (8 clips x 32 frames, M = 50 432) replayed from a hipGraph vs
    issued eagerly, both from the same weights."""
    from sais_amd.graph import GraphedStep
    from sais_amd.loss import calcNCELoss, label_columns
    from sais_amd.optim import SGD
    vit, m, protos = _models()
    B, T = 8, 32
    frames = synth.clips(seed=5, B=B, T=T).view(B * T, 3, 224, 224).to(DEV)
    pad = synth.padding_mask([T] * B).to(DEV)
    cols = label_columns(synth.labels(seed=6, B=B), protos, DEV)
    opt = SGD(list(vit.parameters()) + list(m.parameters()) + list(protos.values()), lr=0.1, engines=[vit, m])
    names, lens = [f"v{b}" for b in range(B)], [T] * B

    maps = []

    def step():
        opt.zero_grad()
        reps = vit(frames).view(B, 1, T, 384)
        emb, attn = m(reps, None, lens, None, 'Prototypes', pad, None, None)
        maps[:] = [attn]                                      # under capture: the graph's static output
        loss = calcNCELoss(0, emb, cols, names, protos, None)
        loss.backward()
        opt.step()
        return loss

    vit(frames[:2])
    m._engine(torch.device(DEV, 0))
    snap = [vit.flat.flat.clone(), m.flat.flat.clone()] + [p.detach().clone() for p in protos.values()]

    def restore():
        with torch.no_grad():
            vit.flat.flat.copy_(snap[0])
            m.flat.flat.copy_(snap[1])
            for p, s in zip(protos.values(), snap[2:]):
                p.copy_(s)
        vit.flat.refresh_shadows(vit._t_names)
        m.flat.refresh_shadows(m._t_names())

    graphed = GraphedStep(step, warmup=2)                     # warm-up + capture run steps: weights move
    attn_static = maps[0]
    # the returned attention map must survive REPLAYS: its zero-fill was a captured hipMemsetAsync until round 6, right on the first
    # replay and garbage from the second on (LABNOTES R6.4); rows sum to one (dropout is off in this test's model)
    for _ in range(3):
        restore()
        graphed()
    torch.cuda.synchronize()
    rows = attn_static.detach().float().sum(-1)
    assert torch.isfinite(attn_static).all() and float((rows - 1).abs().max()) <= 1e-4, float((rows - 1).abs().max())
    restore()
    lg = float(graphed())
    torch.cuda.synchronize()
    g_graph = [vit.flat.grad.clone(), m.flat.grad.clone()] + [p.grad.clone() for p in protos.values()]
    w_graph = [vit.flat.flat.clone(), m.flat.flat.clone()]
    restore()
    le = float(step())
    torch.cuda.synchronize()
    g_eager = [vit.flat.grad.clone(), m.flat.grad.clone()] + [p.grad.clone() for p in protos.values()]
    w_eager = [vit.flat.flat.clone(), m.flat.flat.clone()]
    assert lg == le, (lg, le)                                 # the forward has no atomics: bit-identical
    assert np.isfinite(le) and le > 0
    for a, b in zip(g_graph, g_eager):
        assert float(b.abs().max()) > 0
        assert rel_l2(a, b) <= 1e-5
    for a, b in zip(w_graph, w_eager):
        assert rel_l2(a, b) <= 1e-6
    assert rel_l2(w_eager[0], snap[0]) > 0                    # the step did move the weights
