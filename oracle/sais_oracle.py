"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.

fp32, functional (state-dict in, tensors out) restatement of the SAIS hot path.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(sais_amd/) never does and fails loudly when its HIP library is missing.

Parity status: PINNED.  Every function below is checked against golden vectors produced by
running the reference's own modules in the build container (tests/golden/make_golden.py ->
tests/golden/*.npz; test_oracle_vs_golden.py), to <=2e-5 max-abs on activations.

Paths are relative to /root/reference/SAIS/scripts.
"""
import math

import torch
import torch.nn.functional as F

VIT_HEADS = 6
T_HEADS = 4


# --------------------------------------------------------------------------- ViT-S/16
def vit_patch_embed(sd, x):
    """PatchEmbed.forward — dino-main/vision_transformer.py:116-131.  Conv2d(k=16,s=16) over
    non-overlapping patches == one GEMM on the [F*196, 768] patch matrix."""
    Fn, C, H, W = x.shape
    P = 16
    gh, gw = H // P, W // P
    patches = x.reshape(Fn, C, gh, P, gw, P).permute(0, 2, 4, 1, 3, 5).reshape(Fn, gh * gw, C * P * P)
    w = sd["patch_embed.proj.weight"].reshape(-1, C * P * P)
    return patches @ w.t() + sd["patch_embed.proj.bias"]


def vit_prepare_tokens(sd, x):
    """prepare_tokens — vision_transformer.py:196-207 (pos-embed interpolation is the identity
    at 224x224: :177-178; pos_drop has p=0)."""
    tok = vit_patch_embed(sd, x)
    cls = sd["cls_token"].expand(tok.shape[0], -1, -1)
    return torch.cat((cls, tok), dim=1) + sd["pos_embed"]


def vit_attention(sd, pre, xn, heads=VIT_HEADS, want_probs=False):
    """Attention.forward — vision_transformer.py:68-92."""
    Fn, N, D = xn.shape
    hd = D // heads
    qkv = F.linear(xn, sd[pre + "attn.qkv.weight"], sd[pre + "attn.qkv.bias"])
    q, k, v = qkv.reshape(Fn, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    probs = ((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(dim=-1)
    ctx = (probs @ v).transpose(1, 2).reshape(Fn, N, D)
    out = F.linear(ctx, sd[pre + "attn.proj.weight"], sd[pre + "attn.proj.bias"])
    return out, probs, qkv, ctx


def vit_block(sd, i, x, trace=None, dp=None):
    """Block.forward — vision_transformer.py:95-113, LayerNorm eps 1e-6 (vit_small :243-247), exact-erf GELU (Mlp
    :49-65).  dp = None: eval mode (DropPath = identity).  dp = (s_attn [F], s_mlp [F]): train mode, the per-sample
    DropPath factors keep / (1 - p) of the two residual branches (drop_path :27-46: x.div(keep_prob) * floor(keep_prob +
    rand), one draw per sample) — inputs, because the RNG stream is the framework's, not part of the algorithm."""
    pre = f"blocks.{i}."
    D = x.shape[-1]
    xn = F.layer_norm(x, (D,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], 1e-6)
    a, probs, qkv, ctx = vit_attention(sd, pre, xn)
    mid = x + (a if dp is None else a * dp[0].view(-1, 1, 1))
    xn2 = F.layer_norm(mid, (D,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], 1e-6)
    u = F.linear(xn2, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"])
    h = F.gelu(u)
    y = F.linear(h, sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])
    out = mid + (y if dp is None else y * dp[1].view(-1, 1, 1))
    if trace is not None:
        trace.update(norm1=xn, qkv=qkv, attn_ctx=ctx, proj=a, mid=mid, norm2=xn2, fc1=u, gelu=h, probs=probs)
    return out


def vit_forward(sd, x, depth=12, trace=None, droppath=None):
    """VisionTransformer.forward — vision_transformer.py:209-214: CLS row of the final LN.
    droppath: f32 [2 * depth, F] per-sample branch factors (row 2i = attention of block i, 2i + 1 = its MLP) for train mode."""
    t = vit_prepare_tokens(sd, x)
    if trace is not None:
        trace["tokens"] = t
    for i in range(depth):
        bt = {} if (trace is not None and i == 0) else None
        t = vit_block(sd, i, t, bt, dp=None if droppath is None else (droppath[2 * i], droppath[2 * i + 1]))
        if trace is not None:
            trace[f"block{i}"] = t
            if bt:
                trace.update({"b0_" + k: v for k, v in bt.items()})
    t = F.layer_norm(t, (t.shape[-1],), sd["norm.weight"], sd["norm.bias"], 1e-6)
    return t[:, 0]


def vit_last_selfattention(sd, x, depth=12):
    """get_last_selfattention — vision_transformer.py:216-223 -> [F,6,197,197]."""
    t = vit_prepare_tokens(sd, x)
    for i in range(depth - 1):
        t = vit_block(sd, i, t)
    pre = f"blocks.{depth - 1}."
    xn = F.layer_norm(t, (t.shape[-1],), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], 1e-6)
    return vit_attention(sd, pre, xn)[1]


# --------------------------------------------------------------------------- temporal encoder + head
def temporal_prepare(sd, x):
    """prepareInputForTransformer — prepare_model.py:179-195.  x [B,nsnip,T,D].  The reference
    adds the position rows IN PLACE into the caller's tensor; the oracle (and the product) do
    not mutate inputs — values are identical.  Padded frames also receive a
    position row (App. B.3); RGB and flow share CLS / positions (App. B.2)."""
    B, ns, T, D = x.shape
    pos = torch.cat([sd[f"frame_pos_embeddings.{i}"] for i in range(T)], dim=0)       # [T,D]
    x = x + pos.view(1, 1, T, D)
    cls = sd["frame_cls"].view(1, 1, 1, D).expand(B, ns, 1, D)
    return torch.cat((cls, x), dim=2)                                                  # [B,ns,T+1,D]


# ---- ReLU gates.  A training batch of the temporal encoder has millions of ReLU gates (264 rows x 2048 FFN units x 4 layers
# x streams); a gate whose pre-activation lies within the rounding of the arithmetic that produced it is open in one
# correct evaluation and closed in another, and ONE such gate in the CLS row of a short clip moves that clip's gradient
# by per cents.  To test a backward pass to 1e-4 instead of 1e-2 the checker can therefore evaluate THIS restatement with
# the gates another forward took: `with imposed_gates(list_of_bool_tensors) as ig:` makes every ReLU of the temporal path
# (FFN of each layer, then the aggregate ReLU, per stream; the head's ReLU last — the call order of temporal_forward)
# multiply by the next mask instead of thresholding; ig.mismatches lists, per call, how many gates differed.
_GATES = None


class imposed_gates:
    def __init__(self, gates):
        self.gates, self.mismatches = list(gates), []

    def __enter__(self):
        global _GATES
        _GATES = self
        return self

    def __exit__(self, *exc):
        global _GATES
        _GATES = None
        return False


def _relu(t):
    if _GATES is None or not _GATES.gates:
        return F.relu(t)
    gate = _GATES.gates.pop(0).reshape(t.shape)
    _GATES.mismatches.append(int(((t > 0) != gate).sum()))
    return t * gate.to(t.dtype)


def temporal_layer(sd, pre, x, key_pad, heads=T_HEADS, drop=None, p=0.1):
    """torch-1.8 post-norm TransformerEncoderLayer with the README.md:43-48 edit that returns the head-averaged
    attention map.  x [Bn,S,D] (batch-first here; the reference feeds [S,Bn,D] — same math), key_pad bool [Bn,S].
    drop = None: eval mode (dropout = identity).  drop = {attn [Bn,h,S,S], d1 [Bn,S,D], ff [Bn,S,FF], d2 [Bn,S,D]} keep
    masks: train mode (model.train(), train.py:59) of the layer prepare_model.py:75 builds with torch's default
    dropout p = 0.1 — nn.MultiheadAttention drops the softmaxed weights before P v and RETURNS the dropped weights
    (F.multi_head_attention_forward), then src + dropout1(attn), linear2(dropout(relu(linear1))) and src + dropout2(ff).
    The masks are inputs because the RNG stream is the framework's, not part of the algorithm."""
    Bn, S, D = x.shape
    hd = D // heads
    dm = (lambda t, k: t * drop[k].to(t.dtype) / (1.0 - p)) if drop is not None else (lambda t, k: t)
    qkv = F.linear(x, sd[pre + "self_attn.in_proj_weight"], sd[pre + "self_attn.in_proj_bias"])
    q, k, v = qkv.reshape(Bn, S, 3, heads, hd).permute(2, 0, 3, 1, 4)                  # [Bn,h,S,hd]
    scores = (q * hd ** -0.5) @ k.transpose(-2, -1)
    scores = scores.masked_fill(key_pad.view(Bn, 1, 1, S), float("-inf"))
    probs = dm(scores.softmax(dim=-1), "attn")
    ctx = (probs @ v).transpose(1, 2).reshape(Bn, S, D)
    a = F.linear(ctx, sd[pre + "self_attn.out_proj.weight"], sd[pre + "self_attn.out_proj.bias"])
    x = F.layer_norm(x + dm(a, "d1"), (D,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], 1e-5)
    ff = F.linear(dm(_relu(F.linear(x, sd[pre + "linear1.weight"], sd[pre + "linear1.bias"])), "ff"),
                  sd[pre + "linear2.weight"], sd[pre + "linear2.bias"])
    x = F.layer_norm(x + dm(ff, "d2"), (D,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], 1e-5)
    return x, probs.mean(dim=1)


def temporal_aggregate(sd, seq, pad, nlayers=4, trace=None, drop=None, p=0.1):
    """aggregateInputs — prepare_model.py:197-221.  seq [B,ns,S,D], pad bool [B,ns,S].
    Returns (full relu'd sequence [B,ns,S,D], CLS rows [B,ns,D], attn of LAST layer [B*ns,S,S]).
    drop: per-layer keep masks (temporal_layer) for train mode."""
    B, ns, S, D = seq.shape
    x = seq.reshape(B * ns, S, D)
    kp = pad.reshape(B * ns, S)
    attn = None
    for l in range(nlayers):
        x, attn = temporal_layer(sd, f"transEncoderFrame.layers.{l}.", x, kp, drop=None if drop is None else drop[l], p=p)
        if trace is not None:
            trace.append(x)
    full = _relu(x).reshape(B, ns, S, D)
    return full, full[:, :, 0, :], attn


def temporal_forward(sd, x, f, xpad, fpad, modalities="RGB-Flow", nlayers=4, importance=False, trace=None, drop=None,
                     p=0.1, domains=None):
    """fullModel.forward, data_type='reps', encoder 'ViT', task 'Prototypes', self_attention
    — prepare_model.py:246-448.  Tensor inputs -> (emb [B,256], attn [B*ns,S,S]); list inputs
    (test-time augmentation, :331-346) -> (list of embs, attn of version 0).
    drop = {"rgb": [per-layer masks], "flow": [...]}: train mode with these dropout keep masks (temporal_layer).
    domains (multi-domain models, a state dict WITH linearB, two-stream only — :405-414): samples whose domain is not
    'NH_02' go through linearB instead of linear."""
    if isinstance(x, (list, tuple)):
        embs, attn0 = [], None
        for v in range(len(x)):
            e, a = temporal_forward(sd, x[v], f[v] if f is not None else None, xpad[v],
                                    fpad[v] if fpad is not None else None, modalities, nlayers, domains=domains)[:2]
            embs.append(e)
            attn0 = a if v == 0 else attn0
        return embs, attn0
    full = None
    if modalities in ("RGB", "RGB-Flow"):
        full, rgb, attn = temporal_aggregate(sd, temporal_prepare(sd, x), xpad, nlayers, trace,
                                             drop=None if drop is None else drop.get("rgb"), p=p)
        rep = rgb.mean(dim=1)
    if modalities in ("Flow", "RGB-Flow"):
        ffull, flow, fattn = temporal_aggregate(sd, temporal_prepare(sd, f), fpad, nlayers,
                                                drop=None if drop is None else drop.get("flow"), p=p)
        if modalities == "Flow":
            rep, attn, full = flow.mean(dim=1), fattn, ffull
        else:
            rep = rep + flow.mean(dim=1)                                               # :412
    rrep = _relu(rep)
    emb = F.linear(rrep, sd["linear.weight"], sd["linear.bias"])                       # :416 (double ReLU, App. B.4)
    if "linearB.weight" in sd and modalities == "RGB-Flow" and domains is not None:      # :405-414
        embB = F.linear(rrep, sd["linearB.weight"], sd["linearB.bias"])
        useB = torch.tensor([d != 'NH_02' for d in domains]).view(-1, 1)
        emb = torch.where(useB, embB, emb)
    if importance:
        imp = F.linear(full, sd["importance_function.weight"], sd["importance_function.bias"])   # :419-421
        return imp, emb, attn
    return emb, attn


def mil_forward(sd, x, f, xpad, fpad, nclasses=2, nlayers=4):
    """fullModel.forward, task 'MIL', modalities 'RGB-Flow', eval mode — prepare_model.py:356-361 with getClipReps
    :452-468 and MIL_Head :470-488 / calcAttention :131-138 / obtainVideoRep :140-143 / obtainVideoScore :145-148.
    Per-snippet relu'd CLS rows [B,ns,D] -> + clip position rows -> transEncoderClip (no mask, no CLS token) -> ReLU ->
    per class gated attention over the snippets and a linear score.  Returns the reference's four outputs:
    snip_sequence [ns,B,D] (the position-added encoder INPUT, permuted as :463 leaves it), snip_reps [B,ns,D],
    logits [B,nclasses], {class: attention [B,ns]}.  (The flow stream's clip representations are computed and dropped
    there: MIL_Head(snip_reps, flow_reps=None).)"""
    _, rgb, _ = temporal_aggregate(sd, temporal_prepare(sd, x), xpad, nlayers)
    B, ns, D = rgb.shape
    pos = torch.cat([sd[f"clip_pos_embeddings.{i}"] for i in range(ns)], 0)             # [ns, D]
    seq = rgb + pos.view(1, ns, D)
    z = seq
    for l in range(nlayers):
        z, _ = temporal_layer(sd, f"transEncoderClip.layers.{l}.", z, torch.zeros(B, ns, dtype=torch.bool))
    reps = F.relu(z)
    a = torch.tanh(F.linear(reps, sd["attentionA.weight"], sd["attentionA.bias"]))
    g = torch.sigmoid(F.linear(reps, sd["attentionB.weight"], sd["attentionB.bias"]))
    logits, att = [], {}
    for c in range(nclasses):
        w = torch.softmax(F.linear(a * g, sd[f"attentionModules.{c}.weight"], sd[f"attentionModules.{c}.bias"]), 1)   # [B,ns,1]
        video = (w * reps).sum(1)                                                        # bmm(attention, snip_reps)
        logits.append(F.linear(video, sd[f"finalModules.{c}.weight"], sd[f"finalModules.{c}.bias"]))
        att[c] = w.squeeze(-1)
    return seq.permute(1, 0, 2), reps, torch.cat(logits, 1), att


# --------------------------------------------------------------------------- SupCon / prototype head
def proto_matrix(prototypes):
    keys = list(prototypes.keys())
    return torch.cat([prototypes[k].reshape(1, -1) for k in keys], dim=0), keys


def cosine_logits(emb, prototypes):
    """sim = s_hat . p_hat^T — prepare_miscellaneous.py:16-28 (no epsilon in the norms, App. B.10)."""
    p, _ = proto_matrix(prototypes)
    return (emb / emb.norm(dim=1, keepdim=True)) @ (p / p.norm(dim=1, keepdim=True)).t()


def nce_loss(emb, labels, prototypes):
    """calcNCELoss — prepare_miscellaneous.py:14-46: -mean log( exp(sim[i,y_i]) / sum_j exp(sim[i,j]) ),
    column y_i = index of the prototype whose key == str(label)."""
    sim = cosine_logits(emb, prototypes)
    _, keys = proto_matrix(prototypes)
    cols = torch.tensor([keys.index(str(int(l))) for l in labels])
    e = sim.exp()
    return -(e[torch.arange(len(cols)), cols] / e.sum(dim=1)).log().mean()


def importance_loss(output_importances, importances, ipad, labels):
    """calcImportanceLoss — prepare_miscellaneous.py:48-60, quirks kept: the BCE is reduced to its MEAN over every
    (sample, frame) first, that scalar is then broadcast against the inverted padding mask (whose LAST entry is dropped,
    so frame t is masked by slot t, not t+1), and the mean is taken over the low-skill (label 0) rows only
    (NaN when there are none)."""
    out = output_importances[:, :, 1:, 0]                                   # drop the CLS slot
    bce = F.binary_cross_entropy_with_logits(out, importances, reduction='none').mean()
    mask = (~ipad)[:, :, :-1]
    low = torch.nonzero(labels == 0).flatten()
    return (bce * mask)[low, :].mean()


def probs_from_logits(sim):
    """getProbs / calcProbs — prepare_miscellaneous.py:111-126, process_inference_results.py:76-91."""
    e = sim.exp()
    return e / e.sum(dim=1, keepdim=True)


# --------------------------------------------------------------------------- composition
def e2e_forward(vit_sd, t_sd, clips, fclips, pad, modalities, nlayers=4):
    """clips [B,T,3,224,224] -> ViT per frame -> [B,1,T,384] -> temporal encoder -> emb, attn."""
    B, T = clips.shape[:2]
    reps = vit_forward(vit_sd, clips.reshape(B * T, *clips.shape[2:])).reshape(B, 1, T, -1)
    freps = None
    if modalities == "RGB-Flow":
        freps = vit_forward(vit_sd, fclips.reshape(B * T, *fclips.shape[2:])).reshape(B, 1, T, -1)
    emb, attn = temporal_forward(t_sd, reps, freps, pad, pad, modalities, nlayers)
    return reps, emb, attn


def collate_mask(lens):
    """createPaddingMask — prepare_dataset.py:2798-2806: bool [B,1,maxT+1], True = masked key."""
    maxT = max(lens)
    m = torch.zeros(len(lens), 1, maxT + 1, dtype=torch.bool)
    for b, n in enumerate(lens):
        m[b, :, n + 1:] = True
    return m


def flops_per_frame_fwd():
    """GEMM FLOPs (2*MAC) of one ViT-S/16 forward frame."""
    N, D, H = 197, 384, 1536
    blk = 2 * N * D * 3 * D + 2 * 2 * 6 * N * N * 64 + 2 * N * D * D + 2 * 2 * N * D * H
    return 2 * 196 * 768 * D + 12 * blk
