#!/usr/bin/env python3
"""Generate the committed golden vectors by RUNNING THE REFERENCE's own modules.

Runs only in the build container (needs /root/reference); the GPU box never sees the
reference, only the .npz files this script writes next to itself.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Reference entry points exercised (paths relative to /root/reference/SAIS/scripts):
  dino-main/vision_transformer.py:243-247  vit_small -> VisionTransformer.forward :209-214,
                                           get_last_selfattention :216-223
  prepare_model.py:18-101,179-221,246-448  fullModel(...).forward, Prototypes branch
  prepare_miscellaneous.py:14-46           calcNCELoss
  prepare_dataset.py:2798-2899             loadDataloader.createPaddingMask / pad_collate
Shims: timm / torchvision / h5py are stubbed (never touched on this path) and
nn.TransformerEncoder(Layer).forward are monkey-patched to also return the attention map,
which is the README.md:43-48 hand-edit of torch 1.8 restated for torch 2.10.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import synth  # noqa: E402

REF = "/root/reference/SAIS/scripts"


def import_reference():
    for name in ("timm", "torchvision", "h5py", "cv2"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["timm"].create_model = lambda *a, **k: nn.Identity()
    tv = sys.modules["torchvision"]
    tv.transforms = types.ModuleType("torchvision.transforms")
    tv.models = types.ModuleType("torchvision.models")
    sys.modules["torchvision.transforms"] = tv.transforms
    sys.modules["torchvision.models"] = tv.models
    sys.path.insert(0, os.path.join(REF, "dino-main"))
    sys.path.insert(0, REF)

    # README.md:43-48 — torch-1.8 post-norm layer that also returns the attention map
    def layer_forward(self, src, src_mask=None, src_key_padding_mask=None, **kw):
        src2, attn = self.self_attn(src, src, src, attn_mask=src_mask,
                                    key_padding_mask=src_key_padding_mask, need_weights=True)
        src = self.norm1(src + self.dropout1(src2))
        src2 = self.linear2(self.dropout(self.activation(self.linear1(src))))
        src = self.norm2(src + self.dropout2(src2))
        return src, attn

    def enc_forward(self, src, mask=None, src_key_padding_mask=None, **kw):
        out, attn = src, None
        for mod in self.layers:
            out, attn = mod(out, src_mask=mask, src_key_padding_mask=src_key_padding_mask)
        if self.norm is not None:
            out = self.norm(out)
        return out, attn

    nn.TransformerEncoderLayer.forward = layer_forward
    nn.TransformerEncoder.forward = enc_forward

    import vision_transformer as vits
    import prepare_model
    import prepare_miscellaneous
    return vits, prepare_model, prepare_miscellaneous


ROWS = [0, 1, 100, 196]          # token rows sampled from [F,197,*] activations


def samp(t):
    return t.detach()[:, ROWS].contiguous().numpy().astype(np.float32)


def golden_vit(vits, out):
    torch.manual_seed(0)
    model = vits.vit_small(patch_size=16, drop_path_rate=0.1)
    sd = synth.vit_state_dict(seed=0)
    model.load_state_dict(sd, strict=True)
    model.eval()
    x = synth.clips(seed=10, B=1, T=2)[0]                       # [2,3,224,224]
    g = {}
    acts = {}

    blk0 = model.blocks[0]
    hooks = []

    def keep(mod, out_name=None, in_name=None):
        def fn(m, i, o):
            if out_name:
                acts[out_name] = o
            if in_name:
                acts[in_name] = i[0]
            return None
        hooks.append(mod.register_forward_hook(fn))

    keep(model.patch_embed, "patch")
    keep(blk0.norm1, "b0_norm1")
    keep(blk0.attn.qkv, "b0_qkv")
    keep(blk0.attn.proj, "b0_proj", "b0_attn_ctx")
    keep(blk0.norm2, "b0_norm2", "b0_mid")
    keep(blk0.mlp.act, "b0_gelu", "b0_fc1")
    keep(blk0, "block0", "tokens")
    keep(model.blocks[5], "block5")
    keep(model.blocks[11], "block11")

    with torch.no_grad():
        rep = model(x)
    for h in hooks:
        h.remove()
    with torch.no_grad():
        attn = model.get_last_selfattention(x)                  # [2,6,197,197]
    g["rep"] = rep.numpy()
    g["patch_s"] = acts["patch"][:, [0, 1, 99, 195]].numpy()
    for k in ("tokens", "b0_norm1", "b0_qkv", "b0_attn_ctx", "b0_proj", "b0_mid", "b0_norm2",
              "b0_fc1", "b0_gelu", "block0", "block5", "block11"):
        g[k + "_s"] = samp(acts[k])
    g["attn_rows"] = attn[:, :, [0, 57, 196], :].numpy()        # [2,6,3,197]
    g["attn_colsum"] = attn.sum(dim=2).numpy()                  # [2,6,197] checksum over queries
    g["rows"] = np.array(ROWS)

    # gradients: loss = sum(rep * w)  (drop-path is identity in eval; dropout p=0)
    wvec = torch.randn(2, 384, generator=torch.Generator().manual_seed(77))
    model.zero_grad()
    rep = model(x)
    (rep * wvec).sum().backward()
    g["grad_wvec"] = wvec.numpy()
    for name, p in model.named_parameters():
        gr = p.grad
        g["gnorm/" + name] = np.float32(gr.norm().item())
        if gr.dim() <= 1 or name == "cls_token":
            g["grad/" + name] = gr.numpy()
        elif name == "pos_embed":
            g["grad/" + name] = gr[:, ROWS].numpy()
        elif name == "patch_embed.proj.weight":
            g["grad/" + name] = gr[:8].numpy()
        else:
            g["grad/" + name] = gr[:8].numpy()                  # first 8 output rows
    np.savez_compressed(os.path.join(out, "vit.npz"), **g)
    print("vit.npz: rep", rep.shape, "keys", len(g))


def build_full(prepare_model, nclasses, modalities, nlayers=4, importance=False, seed=1):
    m = prepare_model.fullModel('reps', nclasses, 'in_vs_out', 384, 'ViT', modalities=modalities,
                                freeze_encoder_params=True, self_attention=True, importance_loss=importance)
    if nlayers != 4:                                             # config 1: 1-layer variant (the 4 is hard-coded :76)
        m.transEncoderFrame.layers = m.transEncoderFrame.layers[:nlayers]
        m.transEncoderClip.layers = m.transEncoderClip.layers[:nlayers]
    sd = synth.temporal_state_dict(seed=seed, importance=importance, nlayers=nlayers)
    missing = set(dict(m.state_dict()).keys()) ^ set(sd.keys())
    assert not missing, sorted(missing)[:10]
    m.load_state_dict(sd, strict=True)
    m.eval()
    return m


def golden_temporal(prepare_model, misc, out):
    g = {}
    cases = {"T15": [15, 15, 15], "T12r": [12, 7, 10], "T9r": [9, 1, 5, 9], "T32r": [32, 20, 32, 3, 17, 32, 8, 25]}
    for modal in ("RGB", "RGB-Flow"):
        m = build_full(prepare_model, 2, modal)
        for cname, lens in cases.items():
            B, T = len(lens), max(lens)
            x = synth.reps(seed=100 + T, B=B, T=T)
            f = synth.reps(seed=200 + T, B=B, T=T)
            for b, n in enumerate(lens):                        # pad_collate zero-pads (pad_sequence)
                x[b, :, n:] = 0
                f[b, :, n:] = 0
            pad = synth.padding_mask(lens)
            layer_out = []
            hooks = [l.register_forward_hook(lambda mod, i, o: layer_out.append(o[0].detach().clone()))
                     for l in m.transEncoderFrame.layers]
            with torch.no_grad():
                emb, attn = m(x.clone(), f.clone(), lens, lens, 'Prototypes', pad.clone(), pad.clone(), None)
            for h in hooks:
                h.remove()
            key = f"{modal}/{cname}/"
            g[key + "lens"] = np.array(lens)
            g[key + "emb"] = emb.numpy()
            g[key + "attn"] = attn.numpy()
            # RGB stream runs first: 4 layer outputs [S,B,384] -> keep [B,S,384]
            if modal == "RGB":
                for li in range(4):
                    g[key + f"rgb_layer{li}"] = layer_out[li].permute(1, 0, 2).numpy()
    # TTA list path (prepare_model.py:331-346): 3 versions of lengths 15/12/9, attn from version 0
    m = build_full(prepare_model, 2, "RGB-Flow")
    xs, fs, pads, lens_l = [], [], [], []
    for v, T in enumerate((15, 12, 9)):
        lens = [T, T]
        xs.append(synth.reps(seed=300 + v, B=2, T=T))
        fs.append(synth.reps(seed=400 + v, B=2, T=T))
        pads.append(synth.padding_mask(lens))
        lens_l.append(lens)
    with torch.no_grad():
        embs, attn = m([t.clone() for t in xs], [t.clone() for t in fs], lens_l, lens_l, 'Prototypes',
                       [p.clone() for p in pads], [p.clone() for p in pads], None)
    for v in range(3):
        g[f"TTA/emb{v}"] = embs[v].numpy()
    g["TTA/attn"] = attn.numpy()

    # loss / probs / grads for C in {2,3}
    for C in (2, 3):
        m = build_full(prepare_model, C, "RGB-Flow")
        lens = [32, 20, 32, 3, 17, 32, 8, 25]
        B, T = len(lens), 32
        x = synth.reps(seed=500 + C, B=B, T=T)
        f = synth.reps(seed=600 + C, B=B, T=T)
        for b, n in enumerate(lens):
            x[b, :, n:] = 0
            f[b, :, n:] = 0
        pad = synth.padding_mask(lens)
        protos = nn.ParameterDict({k: nn.Parameter(v.clone()) for k, v in synth.prototypes(seed=2, nclasses=C).items()})
        lab = synth.labels(seed=700 + C, B=B, nclasses=C)
        names = [f"vid_{i}" for i in range(B)]
        x_in = x.clone().requires_grad_(True)
        # the reference adds pos embeddings IN PLACE into its input (prepare_model.py:192);
        # feed a non-leaf copy so autograd still yields d loss / d x
        emb, attn = m(x_in * 1.0, f.clone(), lens, lens, 'Prototypes', pad.clone(), pad.clone(), None)
        loss = misc.calcNCELoss(0, emb, lab, names, protos, None)
        loss.backward()
        key = f"loss/C{C}/"
        g[key + "lens"] = np.array(lens)
        g[key + "labels"] = lab.numpy()
        g[key + "emb"] = emb.detach().numpy()
        g[key + "attn"] = attn.detach().numpy()
        g[key + "loss"] = np.float32(loss.item())
        with torch.no_grad():
            p = torch.vstack(list(protos.values()))
            sim = (emb / emb.norm(dim=1, keepdim=True)) @ (p / p.norm(dim=1, keepdim=True)).T
            g[key + "sim"] = sim.numpy()
            g[key + "probs"] = (sim.exp() / sim.exp().sum(1, keepdim=True)).numpy()
        g[key + "grad_x"] = x_in.grad.numpy()
        for k in protos.keys():
            g[key + f"grad_proto{k}"] = protos[k].grad.numpy()
        P = dict(m.named_parameters())
        for n in ("linear.weight", "linear.bias", "frame_cls", "frame_pos_embeddings.0", "frame_pos_embeddings.31",
                  "transEncoderFrame.layers.0.self_attn.in_proj_bias", "transEncoderFrame.layers.3.self_attn.in_proj_bias",
                  "transEncoderFrame.layers.0.norm1.weight", "transEncoderFrame.layers.3.norm2.bias",
                  "transEncoderFrame.layers.1.linear1.bias", "transEncoderFrame.layers.2.linear2.bias",
                  "transEncoderFrame.layers.2.self_attn.out_proj.bias"):
            g[key + "grad/" + n] = P[n].grad.numpy()
        for n in ("transEncoderFrame.layers.0.self_attn.in_proj_weight", "transEncoderFrame.layers.3.linear1.weight",
                  "transEncoderFrame.layers.1.linear2.weight", "transEncoderFrame.layers.2.self_attn.out_proj.weight"):
            g[key + "grad8/" + n] = P[n].grad[:8].numpy()
        g[key + "ngrads"] = np.int64(sum(1 for q in P.values() if q.grad is not None))
        for n, q in P.items():
            if q.grad is not None and not n.startswith("frame_pos_embeddings"):
                g[key + "gnorm/" + n] = np.float32(q.grad.norm().item())
    np.savez_compressed(os.path.join(out, "temporal.npz"), **g)
    print("temporal.npz keys", len(g))


def golden_importance(prepare_model, misc, out):
    """Optional importance head (-il): fullModel(..., importance_loss=True) returns (importances, emb, attn)
    (prepare_model.py:419-421,444-446); calcImportanceLoss (prepare_miscellaneous.py:48-60)."""
    g = {}
    for modal in ("RGB", "RGB-Flow"):
        m = build_full(prepare_model, 2, modal, importance=True, seed=3)
        lens = [9, 6, 9, 4]
        B, T = len(lens), 9
        x = synth.reps(seed=810, B=B, T=T)
        f = synth.reps(seed=811, B=B, T=T)
        for b, n in enumerate(lens):
            x[b, :, n:] = 0
            f[b, :, n:] = 0
        pad = synth.padding_mask(lens)
        protos = nn.ParameterDict({k: nn.Parameter(v.clone()) for k, v in synth.prototypes(seed=2, nclasses=2).items()})
        lab = torch.tensor([0, 1, 0, 0])
        target = (torch.rand(B, 1, T, generator=torch.Generator().manual_seed(812)) > 0.5).float()
        x_in = x.clone().requires_grad_(True)
        imp, emb, attn = m(x_in * 1.0, f.clone(), lens, lens, 'Prototypes', pad.clone(), pad.clone(), None)
        iloss = misc.calcImportanceLoss(imp, target, pad.clone(), lab)
        loss = misc.calcNCELoss(0, emb, lab, [f"v_{i}" for i in range(B)], protos, None) + iloss
        loss.backward()
        key = modal + "/"
        g[key + "lens"] = np.array(lens)
        g[key + "labels"] = lab.numpy()
        g[key + "target"] = target.numpy()
        g[key + "imp"] = imp.detach().numpy()
        g[key + "emb"] = emb.detach().numpy()
        g[key + "attn"] = attn.detach().numpy()
        g[key + "iloss"] = np.float32(iloss.item())
        g[key + "loss"] = np.float32(loss.item())
        g[key + "grad_x"] = x_in.grad.numpy()
        P = dict(m.named_parameters())
        for n in ("importance_function.weight", "importance_function.bias", "linear.bias", "frame_cls",
                  "transEncoderFrame.layers.3.norm2.bias", "transEncoderFrame.layers.0.self_attn.in_proj_bias"):
            g[key + "grad/" + n] = P[n].grad.numpy()
    # no low-skill sample in the batch: the reference takes the mean of an empty tensor
    imp = torch.randn(2, 1, 5, 1)
    nanloss = misc.calcImportanceLoss(imp, torch.zeros(2, 1, 4), synth.padding_mask([4, 4]), torch.tensor([1, 1]))
    g["empty_low_skill_is_nan"] = np.float32(nanloss.item())
    np.savez_compressed(os.path.join(out, "importance.npz"), **g)
    print("importance.npz keys", len(g))


def collate_batches():
    """Seeded inputs of the two pad_collate branches (shared with tests/test_train_host.py)."""
    lens = [5, 3, 7, 1]
    gen = torch.Generator().manual_seed(5)
    batch = []
    for i, n in enumerate(lens):
        s = torch.randn(1, n, 384, generator=gen)               # nsnippets x nframes x dim
        fl = torch.randn(1, n, 384, generator=gen)
        imp = torch.zeros(1, n)
        batch.append((f"v{i}", s, fl, torch.tensor(i % 2), imp, 'dom'))
    # test-time-augmentation form (val / test / inference items): tuples of 3 versions with different lengths
    tta_lens = [(15, 12, 9), (10, 8, 6), (15, 12, 9)]
    tta_flens = [(2, 2, 1), (1, 1, 1), (2, 1, 1)]
    tta = []
    for i, (ls, fls) in enumerate(zip(tta_lens, tta_flens)):
        xs = tuple(torch.randn(1, n, 384, generator=gen) for n in ls)
        fs = tuple(torch.randn(1, n, 384, generator=gen) for n in fls)
        tta.append((f"w{i}", xs, fs, torch.tensor(0), torch.zeros(1, ls[0]), 'dom'))
    return lens, batch, tta


def golden_collate(out):
    import prepare_dataset
    dl = prepare_dataset.loadDataloader.__new__(prepare_dataset.loadDataloader)
    dl.task = 'Prototypes'
    lens, batch, tta = collate_batches()
    outp = dl.pad_collate(batch)
    names, sp, fp, ip, lab, sl, fl_, sm, fm, im, dom = outp
    g = {"lens": np.array(lens), "snippets_padded": sp.numpy(), "snippets_mask": sm.numpy(),
         "flows_mask": fm.numpy(), "labels": lab.numpy(), "snippets_lens": np.array(sl),
         "flows_padded": fp.numpy(), "flows_lens": np.array(fl_), "importance_padded": ip.numpy(),
         "importance_mask": im.numpy()}
    names, sp, fp, ip, lab, sl, fl_, sm, fm, im, dom = dl.pad_collate(tta)
    assert isinstance(sp, dict) and list(sp.keys()) == [0, 1, 2]
    for v in range(3):
        g[f"tta/snippets_padded_{v}"] = sp[v].numpy()
        g[f"tta/snippets_mask_{v}"] = sm[v].numpy()
        g[f"tta/snippets_lens_{v}"] = np.array(sl[v])
        g[f"tta/flows_padded_{v}"] = fp[v].numpy()
        g[f"tta/flows_mask_{v}"] = fm[v].numpy()
        g[f"tta/flows_lens_{v}"] = np.array(fl_[v])
    g["tta/labels"] = lab.numpy()
    g["tta/importance_padded"] = ip.numpy()
    g["tta/importance_mask"] = im.numpy()
    np.savez_compressed(os.path.join(out, "collate.npz"), **g)
    print("collate.npz", sp[0].shape, sm[0].shape, len(g), "arrays")


def golden_e2e(vits, prepare_model, misc, out):
    """SURVEY §3.4 composition.  (a) config 1: B=1,T=16, 1-layer temporal encoder, RGB;
    (b) B=2,T=4, 4 layers, RGB-Flow, with gradients through ViT + temporal + prototypes."""
    g = {}
    vit = vits.vit_small(patch_size=16, drop_path_rate=0.0)
    vit.load_state_dict(synth.vit_state_dict(seed=0), strict=True)
    vit.eval()

    def run(B, T, nlayers, modal, C, tag, grads):
        m = build_full(prepare_model, C, modal, nlayers=nlayers)
        for p_ in m.parameters():
            p_.grad = None
        vit.zero_grad()
        clips = synth.clips(seed=900 + T, B=B, T=T)
        lens = [T] * B
        pad = synth.padding_mask(lens)
        protos = nn.ParameterDict({k: nn.Parameter(v.clone()) for k, v in synth.prototypes(seed=2, nclasses=C).items()})
        lab = synth.labels(seed=800 + T, B=B, nclasses=C)
        with torch.set_grad_enabled(grads):
            reps = vit(clips.view(B * T, 3, 224, 224)).view(B, 1, T, 384)
            if modal == "RGB-Flow":
                fclips = synth.clips(seed=950 + T, B=B, T=T)
                freps = vit(fclips.view(B * T, 3, 224, 224)).view(B, 1, T, 384)
            else:
                freps = reps.detach().clone()
            emb, attn = m(reps * 1.0, freps * 1.0, lens, lens, 'Prototypes', pad.clone(), pad.clone(), None)
            loss = misc.calcNCELoss(0, emb, lab, [f"v_{i}" for i in range(B)], protos, None)
        p = torch.vstack(list(protos.values())).detach()
        e = emb.detach()
        sim = (e / e.norm(dim=1, keepdim=True)) @ (p / p.norm(dim=1, keepdim=True)).T
        g[tag + "reps"] = reps.detach().numpy()
        g[tag + "emb"] = e.numpy()
        g[tag + "attn"] = attn.detach().numpy()
        g[tag + "sim"] = sim.numpy()
        g[tag + "loss"] = np.float32(loss.item())
        g[tag + "labels"] = lab.numpy()
        if grads:
            loss.backward()
            V = dict(vit.named_parameters())
            for n in ("cls_token", "norm.weight", "blocks.11.mlp.fc2.bias", "blocks.0.attn.qkv.bias",
                      "blocks.6.norm1.weight", "patch_embed.proj.bias"):
                g[tag + "vgrad/" + n] = V[n].grad.numpy()
            for n in ("blocks.0.attn.qkv.weight", "blocks.11.mlp.fc1.weight", "blocks.5.attn.proj.weight"):
                g[tag + "vgrad8/" + n] = V[n].grad[:8].numpy()
            for n, q in V.items():
                g[tag + "vgnorm/" + n] = np.float32(q.grad.norm().item())
            P = dict(m.named_parameters())
            for n in ("linear.bias", "frame_cls", "frame_pos_embeddings.0",
                      "transEncoderFrame.layers.3.norm2.bias", "transEncoderFrame.layers.0.self_attn.in_proj_bias"):
                g[tag + "tgrad/" + n] = P[n].grad.numpy()
            for k in protos.keys():
                g[tag + f"grad_proto{k}"] = protos[k].grad.numpy()

    run(1, 16, 1, "RGB", 2, "cfg1/", grads=False)
    run(2, 4, 4, "RGB-Flow", 2, "train/", grads=True)
    np.savez_compressed(os.path.join(out, "e2e.npz"), **g)
    print("e2e.npz keys", len(g))


def snippet_inputs():
    """nsnippets = 3 (prepare_model.py:179-221,381-382): x, f [B, 3, T, 384], mask [B, 3, T+1] with a different length per
    (clip, snippet); shared with tests/."""
    B, NS, T = 2, 3, 6
    lens = [[6, 4, 5], [3, 6, 1]]
    g = torch.Generator().manual_seed(321)
    x = torch.randn(B, NS, T, 384, generator=g)
    f = torch.randn(B, NS, T, 384, generator=g)
    pad = torch.zeros(B, NS, T + 1, dtype=torch.bool)
    for b in range(B):
        for s_ in range(NS):
            x[b, s_, lens[b][s_]:] = 0
            f[b, s_, lens[b][s_]:] = 0
            pad[b, s_, lens[b][s_] + 1:] = True
    return x, f, pad, synth.labels(seed=77, B=B)


def golden_snippets(prepare_model, misc, out):
    g = {}
    x, f, pad, lab = snippet_inputs()
    for modal in ("RGB", "RGB-Flow"):
        m = build_full(prepare_model, 2, modal)
        protos = nn.ParameterDict({k: nn.Parameter(v.clone()) for k, v in synth.prototypes(seed=2, nclasses=2).items()})
        x_in = x.clone().requires_grad_(True)
        f_in = f.clone().requires_grad_(True)
        emb, attn = m(x_in * 1.0, f_in * 1.0, None, None, 'Prototypes', pad.clone(), pad.clone(), None)
        loss = misc.calcNCELoss(0, emb, lab, ["a", "b"], protos, None)
        loss.backward()
        key = modal + "/"
        g[key + "emb"] = emb.detach().numpy()
        g[key + "attn"] = attn.detach().numpy()
        g[key + "loss"] = np.float32(loss.item())
        g[key + "grad_x"] = x_in.grad.numpy()
        if modal == "RGB-Flow":
            g[key + "grad_f"] = f_in.grad.numpy()
        P = dict(m.named_parameters())
        for n in ("linear.weight", "linear.bias", "frame_cls", "frame_pos_embeddings.0", "frame_pos_embeddings.5",
                  "transEncoderFrame.layers.0.self_attn.in_proj_bias", "transEncoderFrame.layers.3.norm2.bias"):
            g[key + "grad/" + n] = P[n].grad.numpy()
    np.savez_compressed(os.path.join(out, "snippets.npz"), **g)
    print("snippets.npz keys", len(g), "attn", attn.shape)


def multidomain_inputs():
    """Two-domain batch for the per-sample linear / linearB head (prepare_model.py:405-414): shared with tests/."""
    B, T = 4, 7
    lens = [7, 5, 7, 3]
    # seeds chosen so that no pre-ReLU CLS element of either stream lies within 1e-3 of zero (seed 611 puts one at +2e-6: a
    # ReLU gate that any equally valid fp32 summation order can close, which changes that clip's gradient by 10 %)
    x, f = synth.reps(seed=610, B=B, T=T), synth.reps(seed=613, B=B, T=T)
    for b in range(B):
        x[b, :, lens[b]:] = 0
        f[b, :, lens[b]:] = 0
    return x, f, synth.padding_mask(lens), synth.labels(seed=612, B=B), ['NH_02', 'HMH_01', 'HMH_01', 'NH_02'], lens


def golden_multidomain(prepare_model, misc, out):
    """fullModel(domain = 'NH_02+HMH_01', modalities = 'RGB-Flow'): samples whose domain is not 'NH_02' go through linearB."""
    g = {}
    x, f, pad, lab, domains, lens = multidomain_inputs()
    m = prepare_model.fullModel('reps', 2, 'NH_02+HMH_01', 384, 'ViT', modalities='RGB-Flow', freeze_encoder_params=True,
                                self_attention=True, importance_loss=False)
    m.load_state_dict(synth.temporal_state_dict(seed=1, multidomain=True), strict=True)
    m.eval()
    protos = nn.ParameterDict({k: nn.Parameter(v.clone()) for k, v in synth.prototypes(seed=2, nclasses=2).items()})
    x_in, f_in = x.clone().requires_grad_(True), f.clone().requires_grad_(True)
    emb, attn = m(x_in * 1.0, f_in * 1.0, lens, lens, 'Prototypes', pad.clone(), pad.clone(), domains)
    loss = misc.calcNCELoss(0, emb, lab, list("abcd"), protos, domains)
    loss.backward()
    g["emb"], g["attn"], g["loss"] = emb.detach().numpy(), attn.detach().numpy(), np.float32(loss.item())
    g["grad_x"], g["grad_f"] = x_in.grad.numpy(), f_in.grad.numpy()
    P = dict(m.named_parameters())
    for n in ("linear.weight", "linear.bias", "linearB.weight", "linearB.bias", "frame_cls",
              "transEncoderFrame.layers.3.norm2.bias"):
        g["grad/" + n] = P[n].grad.numpy()
    # TTA list form (:405-407): three versions, same per-sample selection
    xs = [x[:, :, :7], x[:, :, :5], x[:, :, :3]]
    fs = [f[:, :, :7], f[:, :, :5], f[:, :, :3]]
    pads = [synth.padding_mask([min(l, n) for l in lens])[:, :, :n + 1] for n in (7, 5, 3)]
    vlens = [[min(l, n) for l in lens] for n in (7, 5, 3)]
    with torch.no_grad():
        embs, attn0 = m([t.clone() for t in xs], [t.clone() for t in fs], vlens, vlens, 'Prototypes',
                        [p_.clone() for p_ in pads], [p_.clone() for p_ in pads], domains)
    for v in range(3):
        g[f"tta/emb{v}"] = embs[v].numpy()
    np.savez_compressed(os.path.join(out, "multidomain.npz"), **g)
    print("multidomain.npz keys", len(g))


def golden_mil(prepare_model, misc, out):
    """task = 'MIL' in eval mode (prepare_model.py:356-361,452-488,131-148): clip-level encoder over the snippet
    representations, gated-attention MIL head.  (Training this path raises inside the reference: DESIGN.md §7.)"""
    g = {}
    x, f, pad, lab = snippet_inputs()
    m = build_full(prepare_model, 2, "RGB-Flow")
    with torch.no_grad():
        seq, reps, logits, att = m(x.clone(), f.clone(), None, None, 'MIL', pad.clone(), pad.clone(), None)
    g["snip_sequence"], g["snip_reps"], g["logits"] = seq.numpy(), reps.numpy(), logits.numpy()
    for c, a in att.items():
        g[f"attention{c}"] = a.numpy()
    np.savez_compressed(os.path.join(out, "mil.npz"), **g)
    print("mil.npz keys", len(g), "seq", tuple(seq.shape), "reps", tuple(reps.shape), "logits", tuple(logits.shape))


def golden_outlier(vits, prepare_model, misc, out):
    """DINO-like dynamic range (synth.vit_state_dict_outlier): ViT features, last-block residual statistics, embeddings,
    attention map and cosine logits of a 2-clip x 8-frame batch, from the reference's own modules."""
    g = {}
    vit = vits.vit_small(patch_size=16, drop_path_rate=0.0)
    vit.load_state_dict(synth.vit_state_dict_outlier(seed=3), strict=True)
    vit.eval()
    B, T = 2, 8
    m = build_full(prepare_model, 2, "RGB", nlayers=4)
    clips = synth.clips(seed=977, B=B, T=T)
    pad = synth.padding_mask([T, T - 3])
    protos = synth.prototypes(seed=2, nclasses=2)
    with torch.no_grad():
        x = vit.prepare_tokens(clips.view(B * T, 3, 224, 224))
        absmax = []
        for blk in vit.blocks:
            x = blk(x)
            absmax.append(x.abs().max().item())
        reps = vit.norm(x)[:, 0].view(B, 1, T, 384)
        emb, attn = m(reps * 1.0, reps * 1.0, [T, T - 3], [T, T - 3], 'Prototypes', pad.clone(), pad.clone(), None)
    p = torch.vstack(list(protos.values()))
    sim = (emb / emb.norm(dim=1, keepdim=True)) @ (p / p.norm(dim=1, keepdim=True)).T
    g["reps"] = reps.numpy()
    g["emb"] = emb.numpy()
    g["attn"] = attn.numpy()
    g["sim"] = sim.numpy()
    g["resid_absmax_per_block"] = np.array(absmax, np.float32)
    np.savez_compressed(os.path.join(out, "outlier.npz"), **g)
    print("outlier.npz: residual |x|max per block", [round(a, 1) for a in absmax], "sim", sim.numpy().round(4).tolist())


DROPOUT_CASE = dict(lens=[9, 4, 7, 9], T=9, C=2, mask_seed=(900, 901), x_seed=(910, 911), label_seed=912, p=0.1)
DROPOUT_GRADS = ("linear.weight", "frame_cls", "frame_pos_embeddings.0", "frame_pos_embeddings.8",
                 "transEncoderFrame.layers.0.self_attn.in_proj_weight", "transEncoderFrame.layers.0.self_attn.in_proj_bias",
                 "transEncoderFrame.layers.1.self_attn.out_proj.weight", "transEncoderFrame.layers.2.linear1.weight",
                 "transEncoderFrame.layers.2.linear1.bias", "transEncoderFrame.layers.3.linear2.weight",
                 "transEncoderFrame.layers.3.norm1.weight", "transEncoderFrame.layers.0.norm2.bias")


def golden_dropout(prepare_model, misc, out):
    """The reference's fullModel in TRAIN mode (train.py:59) with the dropout masks of synth.dropout_masks injected in
    place of torch's RNG: torch.nn.functional.dropout is replaced by a function that pops the next mask, in the order
    the layer calls it (attention weights [Bn*4,S,S], dropout1 [S,Bn,384], dropout [S,Bn,2048], dropout2 [S,Bn,384]; RGB
    stream first, then flow).  Pins WHERE the oracle applies the masks and the 1/(1-p) scaling, forward and backward."""
    import torch.nn.functional as Fn
    c = DROPOUT_CASE
    lens, T, C, p = c["lens"], c["T"], c["C"], c["p"]
    B, S = len(lens), T + 1
    m = build_full(prepare_model, C, "RGB-Flow")
    m.train()
    x, f = synth.reps(seed=c["x_seed"][0], B=B, T=T), synth.reps(seed=c["x_seed"][1], B=B, T=T)
    for b, n in enumerate(lens):
        x[b, :, n:] = 0
        f[b, :, n:] = 0
    pad = synth.padding_mask(lens)
    queue = []
    for sidx in range(2):
        for lm in synth.dropout_masks(c["mask_seed"][sidx], B, S, p=p):
            queue += [lm["attn"].reshape(B * 4, S, S), lm["d1"].transpose(0, 1), lm["ff"].transpose(0, 1), lm["d2"].transpose(0, 1)]
    real = Fn.dropout

    def fed(inp, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return inp
        mask = queue.pop(0)
        assert tuple(mask.shape) == tuple(inp.shape), (tuple(mask.shape), tuple(inp.shape))
        assert abs(p - c["p"]) < 1e-12
        return inp * mask.to(inp.dtype) / (1.0 - p)

    Fn.dropout = fed
    try:
        protos = nn.ParameterDict({k: nn.Parameter(v.clone()) for k, v in synth.prototypes(2, C).items()})
        lab = synth.labels(seed=c["label_seed"], B=B, nclasses=C)
        emb, attn = m(x.clone(), f.clone(), lens, lens, 'Prototypes', pad.clone(), pad.clone(), None)
        loss = misc.calcNCELoss(0, emb, lab, [f"v{b}" for b in range(B)], protos, None)
        loss.backward()
    finally:
        Fn.dropout = real
    assert not queue, len(queue)                                 # 2 streams x 4 layers x 4 sites, all consumed
    g = {"emb": emb.detach().numpy(), "attn": attn.detach().numpy(), "loss": np.array(loss.item()),
         "lens": np.array(lens), "labels": lab.numpy()}
    P = dict(m.named_parameters())
    for n in DROPOUT_GRADS:                                      # big matrices: every 97th element + the norm
        gr = P[n].grad
        g["grad/" + n] = (gr.flatten()[::97] if gr.numel() > 20000 else gr).numpy()
        g["gnorm/" + n] = np.array(gr.norm().item())
    for k, v in protos.items():
        g["grad/proto" + k] = v.grad.numpy()
    # eval-mode outputs of the same inputs: the masks must have changed something
    m.eval()
    with torch.no_grad():
        e0, _ = m(x.clone(), f.clone(), lens, lens, 'Prototypes', pad.clone(), pad.clone(), None)
    g["emb_eval"] = e0.numpy()
    np.savez_compressed(os.path.join(out, "dropout.npz"), **g)
    print("dropout.npz: loss", loss.item(), "|emb - emb_eval|max", float((emb.detach() - e0).abs().max()))


DROPPATH_CASE = dict(frames=6, depth=4, rate=0.3, mask_seed=930, x_seed=931)
DROPPATH_GRADS = ("cls_token", "patch_embed.proj.bias", "blocks.0.attn.qkv.bias", "blocks.1.norm1.weight",
                  "blocks.1.attn.proj.weight", "blocks.2.mlp.fc1.bias", "blocks.2.norm2.bias", "blocks.3.mlp.fc2.weight",
                  "blocks.3.attn.qkv.weight", "norm.weight")


def golden_droppath(vits, out):
    """The reference ViT in TRAIN mode with DropPath (vision_transformer.py:27-46) fed from synth.droppath_factors: the
    module-level drop_path() is replaced by a function that pops the next per-sample factor vector, in call order
    (block 1 attention, block 1 MLP, block 2 ...: block 0 has rate 0 and an nn.Identity).  A 4-block ViT with rate 0.3 so
    that several frames are actually dropped; loss = sum(features * fixed weights)."""
    c = DROPPATH_CASE
    Fn, depth, rate = c["frames"], c["depth"], c["rate"]
    from functools import partial
    # vit_small (:243-247) with fewer blocks: its own constructor call pins depth = 12
    model = vits.VisionTransformer(patch_size=16, embed_dim=384, depth=depth, num_heads=6, mlp_ratio=4, qkv_bias=True,
                                   norm_layer=partial(nn.LayerNorm, eps=1e-6), drop_path_rate=rate)
    model.load_state_dict(synth.vit_state_dict(seed=0, depth=depth), strict=True)
    model.train()
    fac = synth.droppath_factors(c["mask_seed"], Fn, depth, rate)
    queue = [fac[j] for j in range(2, 2 * depth)]                   # rows 0, 1 belong to block 0 (Identity)
    real = vits.drop_path

    def fed(x, drop_prob=0., training=False):
        if drop_prob == 0. or not training:
            return x
        f = queue.pop(0)
        assert set((f * (1 - drop_prob)).round().tolist()) <= {0.0, 1.0}
        return x * f.view(-1, *([1] * (x.ndim - 1)))

    vits.drop_path = fed
    try:
        x = synth.clips(seed=c["x_seed"], B=1, T=Fn)[0]
        w = synth.reps(seed=c["x_seed"] + 1, B=1, T=Fn)[0, 0]        # [Fn, 384] fixed loss weights
        feat = model(x)
        (feat * w).sum().backward()
    finally:
        vits.drop_path = real
    assert not queue
    g = {"feat": feat.detach().numpy(), "factors": fac.numpy()}
    P = dict(model.named_parameters())
    for n in DROPPATH_GRADS:
        gr = P[n].grad
        g["grad/" + n] = (gr.flatten()[::97] if gr.numel() > 20000 else gr).numpy()
        g["gnorm/" + n] = np.array(gr.norm().item())
    model.eval()
    with torch.no_grad():
        g["feat_eval"] = model(x).numpy()
    np.savez_compressed(os.path.join(out, "droppath.npz"), **g)
    print("droppath.npz: dropped", int((fac == 0).sum()), "of", fac.numel(), " |feat - feat_eval|max",
          float(np.abs(g["feat"] - g["feat_eval"]).max()))


def main():
    torch.set_num_threads(8)
    vits, prepare_model, misc = import_reference()
    if len(sys.argv) > 2 and sys.argv[1] == "--only":
        {"dropout": lambda: golden_dropout(prepare_model, misc, HERE),
         "droppath": lambda: golden_droppath(vits, HERE),
         "multidomain": lambda: golden_multidomain(prepare_model, misc, HERE),
         "mil": lambda: golden_mil(prepare_model, misc, HERE)}[sys.argv[2]]()
        return
    golden_dropout(prepare_model, misc, HERE)
    golden_droppath(vits, HERE)
    golden_vit(vits, HERE)
    golden_temporal(prepare_model, misc, HERE)
    golden_collate(HERE)
    golden_e2e(vits, prepare_model, misc, HERE)
    golden_importance(prepare_model, misc, HERE)
    golden_outlier(vits, prepare_model, misc, HERE)
    golden_snippets(prepare_model, misc, HERE)
    golden_multidomain(prepare_model, misc, HERE)
    golden_mil(prepare_model, misc, HERE)


if __name__ == "__main__":
    main()
