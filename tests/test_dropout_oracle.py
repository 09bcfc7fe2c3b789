"""CPU: the oracle's train-mode dropout (masks as inputs) against the reference's fullModel in train() mode with the same
masks injected into torch's dropout calls (tests/golden/make_golden.py::golden_dropout -> dropout.npz)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, os.path.dirname(HERE))
import synth  # noqa: E402
from oracle import sais_oracle as O  # noqa: E402

CASE = dict(lens=[9, 4, 7, 9], T=9, C=2, mask_seed=(900, 901), x_seed=(910, 911), label_seed=912, p=0.1)


def case_inputs():
    c = CASE
    lens, T = c["lens"], c["T"]
    B = len(lens)
    x, f = synth.reps(seed=c["x_seed"][0], B=B, T=T), synth.reps(seed=c["x_seed"][1], B=B, T=T)
    for b, n in enumerate(lens):
        x[b, :, n:] = 0
        f[b, :, n:] = 0
    drop = {"rgb": synth.dropout_masks(c["mask_seed"][0], B, T + 1, p=c["p"]),
            "flow": synth.dropout_masks(c["mask_seed"][1], B, T + 1, p=c["p"])}
    return x, f, synth.padding_mask(lens), drop, synth.labels(seed=c["label_seed"], B=B, nclasses=c["C"])


def test_oracle_train_mode_dropout_matches_reference():
    g = np.load(os.path.join(HERE, "golden", "dropout.npz"))
    x, f, pad, drop, lab = case_inputs()
    assert list(g["lens"]) == CASE["lens"] and np.array_equal(g["labels"], lab.numpy())
    sd = {k: v.clone().requires_grad_(True) for k, v in synth.temporal_state_dict(seed=1).items()}
    pr = {k: v.clone().requires_grad_(True) for k, v in synth.prototypes(2, CASE["C"]).items()}
    emb, attn = O.temporal_forward(sd, x, f, pad, pad, "RGB-Flow", drop=drop, p=CASE["p"])
    loss = O.nce_loss(emb, lab, pr)
    loss.backward()
    assert np.abs(emb.detach().numpy() - g["emb"]).max() < 2e-5
    assert np.abs(attn.detach().numpy() - g["attn"]).max() < 1e-6          # the returned map is the DROPPED one
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    assert np.abs(g["emb"] - g["emb_eval"]).max() > 0.5                     # and the masks did change the outputs
    for k in g.files:
        if not k.startswith("grad/"):
            continue
        n = k[5:]
        got = pr[n[5:]].grad if n.startswith("proto") else sd[n].grad
        want = g[k]
        if got.numel() != want.size:
            assert abs(got.norm().item() - float(g["gnorm/" + n])) <= 1e-4 * max(1.0, float(g["gnorm/" + n])), n
            got = got.flatten()[::97]
        err = np.abs(got.numpy().reshape(want.shape) - want).max()
        assert err <= 2e-5 * max(1.0, np.abs(want).max()), (n, err)
    # masks of ones with p = 0 are the eval path
    ones = {s: [{k: torch.ones_like(v) for k, v in lm.items()} for lm in drop[s]] for s in drop}
    with torch.no_grad():
        e1, _ = O.temporal_forward(sd, x, f, pad, pad, "RGB-Flow", drop=ones, p=0.0)
        e0, _ = O.temporal_forward(sd, x, f, pad, pad, "RGB-Flow")
    assert torch.equal(e0, e1) and np.abs(e0.numpy() - g["emb_eval"]).max() < 2e-5


def test_oracle_train_mode_droppath_matches_reference():
    """droppath.npz: the reference ViT (4 blocks, rate 0.3) in train() with synth.droppath_factors injected into its
    drop_path(): features and parameter gradients."""
    g = np.load(os.path.join(HERE, "golden", "droppath.npz"))
    Fn, depth, rate = 6, 4, 0.3
    fac = synth.droppath_factors(930, Fn, depth, rate)
    assert np.array_equal(fac.numpy(), g["factors"]) and float((fac == 0).sum()) >= 5 and bool((fac[:2] == 1).all())
    sd = {k: v.clone().requires_grad_(True) for k, v in synth.vit_state_dict(seed=0, depth=depth).items()}
    x = synth.clips(seed=931, B=1, T=Fn)[0]
    w = synth.reps(seed=932, B=1, T=Fn)[0, 0]
    feat = O.vit_forward(sd, x, depth=depth, droppath=fac)
    (feat * w).sum().backward()
    assert np.abs(feat.detach().numpy() - g["feat"]).max() < 2e-4
    assert np.abs(g["feat"] - g["feat_eval"]).max() > 0.5
    for k in g.files:
        if not k.startswith("grad/"):
            continue
        n = k[5:]
        got, want = sd[n].grad, g[k]
        if got.numel() != want.size:
            assert abs(got.norm().item() - float(g["gnorm/" + n])) <= 1e-3 * max(1.0, float(g["gnorm/" + n])), n
            got = got.flatten()[::97]
        err = np.abs(got.numpy().reshape(want.shape) - want).max()
        assert err <= 1e-3 * max(1.0, np.abs(want).max()), (n, err)
    with torch.no_grad():
        assert np.abs(O.vit_forward(sd, x, depth=depth).numpy() - g["feat_eval"]).max() < 2e-4
