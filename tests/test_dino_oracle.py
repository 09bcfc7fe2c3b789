"""Pins oracle/dino_oracle.py against the vectors the reference itself produced (tests/golden/make_golden_dino.py:
main_dino.DINOLoss / train_one_epoch body, vision_transformer.DINOHead / interpolate_pos_encoding, utils.*,
torch.optim.AdamW).  CPU only."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import synth  # noqa: E402

from oracle import dino_oracle as do  # noqa: E402


@pytest.fixture(scope="module")
def gstep():
    return np.load(os.path.join(HERE, "golden", "dino_step.npz"))


@pytest.fixture(scope="module")
def cfg(gstep):
    return {k: float(v) for k, v in zip(gstep["cfg_keys"], gstep["cfg_vals"])}


def sample(t):
    t = t.detach().reshape(-1)
    return (t[::97] if t.numel() > 20000 else t).double().numpy()


def student_sd(out_dim):
    sd = {"backbone." + k: v for k, v in synth.vit_state_dict(seed=20).items()}
    sd.update({"head." + k: v for k, v in synth.dino_head_state_dict(seed=21, out_dim=out_dim).items()})
    return sd


def test_schedules(gstep, cfg):
    c = cfg
    lr = do.cosine_scheduler(c["lr"] * c["B"] / 256.0, c["min_lr"], int(c["epochs"]), int(c["niter_per_ep"]),
                             warmup_epochs=int(c["warmup_epochs"]))
    wd = do.cosine_scheduler(c["weight_decay"], c["weight_decay_end"], int(c["epochs"]), int(c["niter_per_ep"]))
    mom = do.cosine_scheduler(c["momentum_teacher"], 1, int(c["epochs"]), int(c["niter_per_ep"]))
    tt = do.teacher_temp_schedule(c["warmup_teacher_temp"], c["teacher_temp"], int(c["warmup_teacher_temp_epochs"]),
                                  int(c["epochs"]))
    for a, b in ((lr, "lr_schedule"), (wd, "wd_schedule"), (mom, "momentum_schedule"), (tt, "teacher_temp_schedule")):
        assert a.shape == gstep[b].shape
        np.testing.assert_allclose(a, gstep[b], rtol=0, atol=1e-15)


def test_param_groups(gstep):
    sd = student_sd(16)
    names = list(gstep["param_names"])
    assert names == list(sd.keys())                                  # named_parameters() order of MultiCropWrapper
    reg = [do.is_regularized(n, sd[n].shape) and bool(r) for n, r in zip(names, gstep["requires_grad"])]
    assert reg == [bool(x) for x in gstep["regularized"]]
    assert [n for n, r in zip(names, gstep["requires_grad"]) if not r] == ["head.last_layer.weight_g"]


def test_pos_interpolation(gstep):
    pos = synth.vit_state_dict(seed=20)["pos_embed"].double()
    got = do.interpolate_pos_encoding(pos, 36, 96, 96)[0].numpy()
    assert got.shape == (37, 384)
    assert np.abs(got - gstep["pos_embed_96"]).max() < 2e-6          # the golden is fp32
    W = do.pos_interp_matrix(14, 96, 96)
    assert W.shape == (36, 196) and np.allclose(W.sum(1), 1.0)


def test_backbone_at_96(gstep):
    bb = {k: v.double() for k, v in synth.vit_state_dict(seed=20).items()}
    x = synth.dino_crops(seed=300, B=2, n_local=1)[2].double()
    got = do.vit_forward_res(bb, x).numpy()
    assert np.abs(got - gstep["cls_96"]).max() < 2e-5


def test_dino_loss_65536():
    g = np.load(os.path.join(HERE, "golden", "dino_loss.npz"))
    n, B, ncrops = 65536, 2, 10
    gen = synth._gen(410)
    s0 = torch.randn(ncrops * B, n, generator=gen) * 0.3
    t = (torch.randn(2 * B, n, generator=gen) * 0.3).double()
    c0 = (torch.randn(1, n, generator=gen) * 0.05).double()
    sched = do.teacher_temp_schedule(0.04, 0.07, 3, 10)
    for epoch in (0, 5):
        s = s0.double().requires_grad_(True)
        loss = do.dino_loss(s, t, c0, float(sched[epoch]), ncrops)
        loss.backward()
        assert abs(float(loss.detach()) - float(g[f"e{epoch}/loss"])) < 2e-5 * 15.5        # the golden is an fp32 sum
        ref = g[f"e{epoch}/grad_cols"]
        assert np.abs(s.grad[:, ::257].numpy() - ref).max() < 1e-5 * np.abs(ref).max()
        np.testing.assert_allclose(s.grad.sum(1).numpy(), g[f"e{epoch}/grad_rowsum"], atol=5e-6)
        np.testing.assert_allclose(s.grad.abs().sum(1).numpy(), g[f"e{epoch}/grad_abs_sum"], rtol=2e-5)
        c1 = do.center_update(c0, t)
        assert np.abs(c1.numpy() - g[f"e{epoch}/center_after"]).max() < 1e-7


def test_train_steps(gstep, cfg):
    """Four iterations of train_one_epoch's body: losses, outputs, centre, per-parameter norms (clipping active on
    about half of the tensors), frozen last layer in epoch 0, AdamW on two groups, EMA teacher."""
    c = cfg
    out_dim, n_local = int(c["out_dim"]), int(c["n_local"])
    st = do.TrainState(student_sd(out_dim))
    track = sorted({k.split("/", 2)[2] for k in gstep.files if k.startswith("it0/student/")})
    for it in range(int(c["iters"])):
        epoch = it // int(c["niter_per_ep"])
        crops = synth.dino_crops(seed=300 + it, B=int(c["B"]), n_local=n_local)
        r = do.train_step(st, crops, it, epoch, gstep["lr_schedule"], gstep["wd_schedule"], gstep["momentum_schedule"],
                          gstep["teacher_temp_schedule"], c["clip_grad"], int(c["freeze_last_layer"]), n_local)
        k = f"it{it}/"
        assert abs(r["loss"] - float(gstep[k + "loss"])) < 2e-5, it
        assert np.abs(r["teacher_out"].numpy() - gstep[k + "teacher_out"]).max() < 3e-5
        assert np.abs(r["student_out"].numpy() - gstep[k + "student_out"]).max() < 3e-5
        assert np.abs(r["center_before"].numpy() - gstep[k + "center_before"]).max() < 1e-6
        assert np.abs(st.center.numpy() - gstep[k + "center_after"]).max() < 1e-6
        names = list(gstep[k + "norm_names"])
        got = np.array([r["norms"][n] for n in names])
        np.testing.assert_allclose(got, gstep[k + "norms"], rtol=2e-3, atol=1e-7)
        if it == 0:
            assert np.abs(r["dlogits"].numpy() - gstep["dlogits0"]).max() < 3e-5 * np.abs(gstep["dlogits0"]).max()
            for n in track:
                if "grad0/" + n in gstep.files:
                    ref = gstep["grad0/" + n]
                    err = np.linalg.norm(sample(r["grads"][n]) - ref) / max(np.linalg.norm(ref), 1e-30)
                    assert err < 2e-4, (n, err)
        for n in track:
            for who, sd in (("student", st.student), ("teacher", st.teacher)):
                ref = gstep[k + who + "/" + n]
                # Adam's first steps move every element by ~lr whatever the gradient's size: elements whose fp32
                # gradient is pure rounding noise may differ by up to 2 lr from the fp64 oracle
                assert np.abs(sample(sd[n]) - ref).max() < 2.5 * 2e-4 + 1e-6, (it, who, n)
                assert np.median(np.abs(sample(sd[n]) - ref)) < 2e-6, (it, who, n)
    assert st.steps["head.last_layer.weight_v"] == 2 and st.steps["backbone.cls_token"] == 4
