"""Deterministic synthetic weights / inputs shared by the golden-vector generator
(tests/golden/make_golden.py, which imports the reference in the build container) and by
the tests / smoke / bench that replay the same inputs on the GPU box, where the reference
does not exist.  Everything is drawn from a seeded CPU torch.Generator so both sides see
bit-identical tensors.

Key lists mirror the reference's state_dict contracts:
  * ViT-S/16:   SAIS/scripts/dino-main/vision_transformer.py:134-172,243-247  (150 tensors)
  * fullModel:  SAIS/scripts/prepare_model.py:46-91  (ViT / reps / Prototypes; no encoder.*)
"""
import torch

VIT_DIM, VIT_DEPTH, VIT_HEADS, VIT_MLP, VIT_TOKENS, PATCH = 384, 12, 6, 1536, 197, 16
T_DIM, T_HEADS, T_FF, T_LAYERS, T_NPOS, EMB = 384, 4, 2048, 4, 2000, 256


def _gen(seed):
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return g


def vit_keys(depth=VIT_DEPTH):
    """(name, shape, kind) in the reference's state_dict order."""
    D = VIT_DIM
    out = [("cls_token", (1, 1, D), "emb"), ("pos_embed", (1, VIT_TOKENS, D), "emb"),
           ("patch_embed.proj.weight", (D, 3, PATCH, PATCH), "w"), ("patch_embed.proj.bias", (D,), "b")]
    for i in range(depth):
        p = f"blocks.{i}."
        out += [(p + "norm1.weight", (D,), "g"), (p + "norm1.bias", (D,), "b"),
                (p + "attn.qkv.weight", (3 * D, D), "w"), (p + "attn.qkv.bias", (3 * D,), "b"),
                (p + "attn.proj.weight", (D, D), "w"), (p + "attn.proj.bias", (D,), "b"),
                (p + "norm2.weight", (D,), "g"), (p + "norm2.bias", (D,), "b"),
                (p + "mlp.fc1.weight", (VIT_MLP, D), "w"), (p + "mlp.fc1.bias", (VIT_MLP,), "b"),
                (p + "mlp.fc2.weight", (D, VIT_MLP), "w"), (p + "mlp.fc2.bias", (D,), "b")]
    out += [("norm.weight", (D,), "g"), ("norm.bias", (D,), "b")]
    return out


def vit_state_dict(seed=0, depth=VIT_DEPTH, w_std=0.04):
    """Random ViT-S/16 weights.  Deliberately NOT the constructor init (zeros biases, unit
    gammas) so that every bias/gamma/beta term is exercised by the parity tests."""
    g = _gen(seed)
    sd = {}
    for name, shape, kind in vit_keys(depth):
        if kind == "w":
            t = torch.randn(shape, generator=g) * w_std
        elif kind == "emb":
            t = torch.randn(shape, generator=g) * 0.05
        elif kind == "g":
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            t = 0.05 * torch.randn(shape, generator=g)
        sd[name] = t.float()
    return sd


OUTLIER_CHANNELS = (7, 101, 300)


def vit_state_dict_outlier(seed=3, depth=VIT_DEPTH):
    """ViT-S/16 weights with the dynamic range of a TRAINED DINO checkpoint rather than of an initialiser:
      * "massive-activation" residual channels (pushed to ~ +-25 by the first blocks' fc2 biases and growing to |x| ~ 70),
        squashed by small LayerNorm gains on those channels, as trained models do;
      * LayerNorm gains of x20 - x50 on four other channels per norm, i.e. |xn| ~ 100 next to O(1) values in one row,
        with the consuming weight columns (qkv / fc1) scaled down so that those channels carry ~4x a normal channel's
        contribution - large gains in trained models pair with small downstream weights; without that the attention
        logits reach |200| (std 20) and ANY bf16 path, torch's own autocast included, is 25 % off;
      * attention logits ~6x sharper (|logit| up to ~30) than the friendly fixture's.
    What the bf16 operand path has to survive: outliers in the residual stream and in the normalised rows."""
    sd = vit_state_dict(seed=seed, depth=depth)
    g = _gen(seed + 1000)
    for i in range(depth):
        p = f"blocks.{i}."
        for c, mag in zip(OUTLIER_CHANNELS, (25.0, -18.0, 30.0)):
            if i < 2:
                sd[p + "mlp.fc2.bias"][c] += mag / 2                 # the residual stream picks the outliers up early
            sd[p + "norm1.weight"][c] *= 0.05                        # ... and trained LN gains squash them again
            sd[p + "norm2.weight"][c] *= 0.05
        for norm, consumer in (("norm1", "attn.qkv.weight"), ("norm2", "mlp.fc1.weight")):
            hot = torch.randint(0, VIT_DIM, (4,), generator=g)
            gain = 20.0 + 30.0 * torch.rand(4, generator=g)          # x20 - x50 gains
            sd[p + norm + ".weight"][hot] *= gain
            sd[p + consumer][:, hot] *= 4.0 / gain
        sd[p + "attn.qkv.weight"][:2 * VIT_DIM] *= 2.5               # sharper attention
    sd["norm.weight"][list(OUTLIER_CHANNELS)] *= 0.05
    return sd


def temporal_keys(importance=False, nlayers=T_LAYERS):
    """fullModel('reps', nclasses, domain, 384, 'ViT') parameter contract,
    without the encoder.* ballast."""
    D = T_DIM
    out = []
    out += [("linear.weight", (EMB, D), "w"), ("linear.bias", (EMB,), "b"),
            ("linear2.weight", (3, EMB), "w"), ("linear2.bias", (3,), "b")]
    if importance:
        out += [("importance_function.weight", (1, D), "w"), ("importance_function.bias", (1,), "b")]
    out += [("frame_cls", (1, D), "u"), ("clip_cls", (1, D), "u")]
    out += [(f"frame_pos_embeddings.{i}", (1, D), "u") for i in range(T_NPOS)]
    out += [(f"clip_pos_embeddings.{i}", (1, D), "u") for i in range(T_NPOS)]
    for enc in ("transEncoderFrame", "transEncoderClip"):
        for l in range(nlayers):
            p = f"{enc}.layers.{l}."
            out += [(p + "self_attn.in_proj_weight", (3 * D, D), "w"), (p + "self_attn.in_proj_bias", (3 * D,), "b"),
                    (p + "self_attn.out_proj.weight", (D, D), "w"), (p + "self_attn.out_proj.bias", (D,), "b"),
                    (p + "linear1.weight", (T_FF, D), "w"), (p + "linear1.bias", (T_FF,), "b"),
                    (p + "linear2.weight", (D, T_FF), "w"), (p + "linear2.bias", (D,), "b"),
                    (p + "norm1.weight", (D,), "g"), (p + "norm1.bias", (D,), "b"),
                    (p + "norm2.weight", (D,), "g"), (p + "norm2.bias", (D,), "b")]
    out += [("attentionA.weight", (EMB, D), "w"), ("attentionA.bias", (EMB,), "b"),
            ("attentionB.weight", (EMB, D), "w"), ("attentionB.bias", (EMB,), "b")]
    for c in range(3):
        out += [(f"attentionModules.{c}.weight", (1, EMB), "w"), (f"attentionModules.{c}.bias", (1,), "b")]
    for c in range(3):
        out += [(f"finalModules.{c}.weight", (1, D), "w"), (f"finalModules.{c}.bias", (1,), "b")]
    return out


def temporal_state_dict(seed=1, importance=False, nlayers=T_LAYERS, w_std=0.05, multidomain=False):
    """Each of the 4 layers gets independent weights (nn.TransformerEncoder clones one layer,
    which would hide layer-indexing bugs: SURVEY §8c).  multidomain: plus linearB ('+' in the domain name,
    prepare_model.py:47-48), drawn AFTER everything else so the other tensors are those of the single-domain dict."""
    g = _gen(seed)
    sd = {}
    keys = temporal_keys(importance, nlayers)
    if multidomain:
        keys = keys + [("linearB.weight", (EMB, T_DIM), "w"), ("linearB.bias", (EMB,), "b")]
    for name, shape, kind in keys:
        if kind == "w":
            t = torch.randn(shape, generator=g) * w_std
        elif kind == "u":
            t = torch.rand(shape, generator=g)          # reference: torch.rand (prepare_model.py:62-68)
        elif kind == "g":
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            t = 0.05 * torch.randn(shape, generator=g)
        sd[name] = t.float()
    return sd


def prototypes(seed=2, nclasses=2):
    g = _gen(seed)
    return {str(c): torch.rand((1, EMB), generator=g) for c in range(nclasses)}   # prepare_model.py:556-560


IMNET_MEAN = (0.485, 0.456, 0.406)
IMNET_STD = (0.229, 0.224, 0.225)


def clips(seed, B, T, hw=224):
    """uint8 pixels -> /255 -> ImageNet normalise (extract_representations.py:148,158-162)."""
    g = _gen(seed)
    u8 = torch.randint(0, 256, (B, T, 3, hw, hw), generator=g, dtype=torch.uint8)
    x = u8.float() / 255.0
    mean = torch.tensor(IMNET_MEAN).view(1, 1, 3, 1, 1)
    std = torch.tensor(IMNET_STD).view(1, 1, 3, 1, 1)
    return (x - mean) / std


def reps(seed, B, T, scale=1.0):
    """Stand-in per-frame ViT reps [B,1,T,384] for the temporal-only goldens."""
    g = _gen(seed)
    return torch.randn((B, 1, T, T_DIM), generator=g) * scale


def labels(seed, B, nclasses=2):
    g = _gen(seed)
    return torch.randint(0, nclasses, (B,), generator=g)


def padding_mask(lens, maxT=None):
    """bool [B,1,maxT+1], True = masked key; CLS (slot 0) never masked
    (prepare_dataset.py:2798-2806)."""
    maxT = max(lens) if maxT is None else maxT
    m = torch.zeros(len(lens), 1, maxT + 1, dtype=torch.bool)
    for b, n in enumerate(lens):
        m[b, :, n + 1:] = True
    return m


def dropout_masks(seed, Bn, S, nlayers=T_LAYERS, p=0.1):
    """Per-layer Bernoulli(1-p) keep masks for the four dropout sites of a train-mode TransformerEncoderLayer, in the
    oracle's batch-first layouts: attn [Bn,4,S,S], d1 [Bn,S,384], ff [Bn,S,2048], d2 [Bn,S,384]."""
    g = _gen(seed)
    k = lambda *sh: torch.rand(sh, generator=g) >= p
    return [dict(attn=k(Bn, T_HEADS, S, S), d1=k(Bn, S, T_DIM), ff=k(Bn, S, T_FF), d2=k(Bn, S, T_DIM)) for _ in range(nlayers)]


def droppath_factors(seed, frames, depth=VIT_DEPTH, rate=0.1):
    """[2 * depth, frames] DropPath factors keep / (1 - p_i) (p_i = linspace(0, rate, depth)[i], vision_transformer.py:150):
    row 2i = attention branch of block i, 2i + 1 = its MLP branch.  Block 0 has p = 0 (always 1)."""
    g = _gen(seed)
    rates = torch.linspace(0, rate, depth).repeat_interleave(2)
    keep = torch.rand((2 * depth, frames), generator=g) >= rates.view(-1, 1)
    return keep.float() / (1.0 - rates).view(-1, 1)


# --------------------------------------------------------------------------- DINO pre-training (main_dino.py)
HEAD_HID, HEAD_BOT = 2048, 256


def dino_head_keys(out_dim):
    """DINOHead(384, out_dim) parameter contract in named_parameters() order (vision_transformer.py:257-291):
    mlp = Linear, GELU, Linear, GELU, Linear; last_layer = weight_norm(Linear(256, out_dim, bias=False))."""
    return [("mlp.0.weight", (HEAD_HID, VIT_DIM), "w"), ("mlp.0.bias", (HEAD_HID,), "b"),
            ("mlp.2.weight", (HEAD_HID, HEAD_HID), "w"), ("mlp.2.bias", (HEAD_HID,), "b"),
            ("mlp.4.weight", (HEAD_BOT, HEAD_HID), "w"), ("mlp.4.bias", (HEAD_BOT,), "b"),
            ("last_layer.weight_g", (out_dim, 1), "one"), ("last_layer.weight_v", (out_dim, HEAD_BOT), "w")]


def dino_head_state_dict(seed, out_dim, w_std=0.04):
    g = _gen(seed)
    sd = {}
    for name, shape, kind in dino_head_keys(out_dim):
        if kind == "w":
            t = torch.randn(shape, generator=g) * w_std
        elif kind == "one":
            t = torch.ones(shape)                           # weight_g.data.fill_(1), :278
        else:
            t = 0.05 * torch.randn(shape, generator=g)
        sd[name] = t.float()
    return sd


def dino_crops(seed, B, n_local, gsize=224, lsize=96):
    """The list DataAugmentationDINO.__call__ returns after collation: 2 global [B,3,224,224] + n_local [B,3,96,96]
    normalised crops (main_dino.py:633-679).  Synthetic pixels: the PIL augmentations are outside the path."""
    g = _gen(seed)
    mean = torch.tensor(IMNET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(IMNET_STD).view(1, 3, 1, 1)
    out = []
    for i in range(2 + n_local):
        s = gsize if i < 2 else lsize
        u8 = torch.randint(0, 256, (B, 3, s, s), generator=g, dtype=torch.uint8)
        out.append((u8.float() / 255.0 - mean) / std)
    return out
