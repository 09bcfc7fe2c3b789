#!/usr/bin/env python3
"""Golden vectors of the DINO pre-training objective, produced by RUNNING THE REFERENCE's own modules
(build container only: needs /root/reference).

    python tests/golden/make_golden_dino.py        # rewrites tests/golden/dino_*.npz

Reference entry points exercised (paths relative to /root/reference/SAIS/scripts/dino-main):
  main_dino.py:517-576    train_one_epoch body: schedules -> teacher(images[:2]) / student(images) -> DINOLoss ->
                          backward -> utils.clip_gradients -> utils.cancel_gradients_last_layer -> AdamW.step -> EMA
  main_dino.py:579-630    DINOLoss.forward / update_center (dist.all_reduce on a 1-rank gloo group)
  vision_transformer.py:174-214  interpolate_pos_encoding (bicubic, 96x96 crops), prepare_tokens, forward
  vision_transformer.py:257-291  DINOHead
  utils.py:132-150,187-198,595-645  clip_gradients, cancel_gradients_last_layer, cosine_scheduler, MultiCropWrapper,
                          get_params_groups
torchvision / timm are stubbed exactly as in make_golden.py (never touched on this path); fp32 (use_fp16 False branch,
main_dino.py:545-552); drop_path_rate 0 (DropPath draws from torch's RNG, pinned separately by droppath.npz).
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

# the fixture's training configuration (every value is an argument of main_dino.get_args_parser)
CFG = dict(B=2, n_local=2, out_dim=1024, iters=4, niter_per_ep=2, epochs=3, warmup_epochs=1, lr=0.0256, min_lr=1e-5,
           weight_decay=0.04, weight_decay_end=0.4, momentum_teacher=0.9, clip_grad=0.02, freeze_last_layer=1,
           warmup_teacher_temp=0.04, teacher_temp=0.07, warmup_teacher_temp_epochs=2)

TRACK = ["backbone.cls_token", "backbone.pos_embed", "backbone.patch_embed.proj.bias", "backbone.patch_embed.proj.weight",
         "backbone.blocks.0.attn.qkv.weight", "backbone.blocks.0.norm1.weight", "backbone.blocks.5.mlp.fc1.weight",
         "backbone.blocks.11.mlp.fc2.bias", "backbone.blocks.11.attn.proj.weight", "backbone.norm.weight",
         "head.mlp.0.bias", "head.mlp.2.weight", "head.mlp.4.weight", "head.last_layer.weight_g",
         "head.last_layer.weight_v"]


def sample(t):
    t = t.detach().reshape(-1)
    return (t[::97] if t.numel() > 20000 else t).numpy().astype(np.float32).copy()


def import_reference():
    for name in ("timm", "torchvision", "h5py", "cv2"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    tv = sys.modules["torchvision"]
    for sub in ("transforms", "models", "datasets"):
        m = types.ModuleType("torchvision." + sub)
        setattr(tv, sub, m)
        sys.modules["torchvision." + sub] = m
    sys.path.insert(0, os.path.join(REF, "dino-main"))
    import main_dino
    import utils
    import vision_transformer as vits
    return main_dino, utils, vits


def init_dist():
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29613", rank=0, world_size=1)


def build(utils, vits, out_dim):
    student = utils.MultiCropWrapper(vits.vit_small(patch_size=16, drop_path_rate=0.0),
                                     vits.DINOHead(384, out_dim, use_bn=False, norm_last_layer=True))
    teacher = utils.MultiCropWrapper(vits.vit_small(patch_size=16), vits.DINOHead(384, out_dim, False))
    sd = {"backbone." + k: v for k, v in synth.vit_state_dict(seed=20).items()}
    sd.update({"head." + k: v for k, v in synth.dino_head_state_dict(seed=21, out_dim=out_dim).items()})
    student.load_state_dict(sd, strict=True)
    teacher.load_state_dict(student.state_dict())                 # main_dino.py:417
    for p in teacher.parameters():
        p.requires_grad = False
    return student, teacher


def golden_step(main_dino, utils, vits, out):
    c = CFG
    torch.manual_seed(0)
    student, teacher = build(utils, vits, c["out_dim"])
    loss_mod = main_dino.DINOLoss(c["out_dim"], c["n_local"] + 2, c["warmup_teacher_temp"], c["teacher_temp"],
                                  c["warmup_teacher_temp_epochs"], c["epochs"])
    groups = utils.get_params_groups(student)
    opt = torch.optim.AdamW(groups)
    lr_s = utils.cosine_scheduler(c["lr"] * c["B"] * 1 / 256.0, c["min_lr"], c["epochs"], c["niter_per_ep"],
                                  warmup_epochs=c["warmup_epochs"])
    wd_s = utils.cosine_scheduler(c["weight_decay"], c["weight_decay_end"], c["epochs"], c["niter_per_ep"])
    mom_s = utils.cosine_scheduler(c["momentum_teacher"], 1, c["epochs"], c["niter_per_ep"])
    g = {"lr_schedule": lr_s, "wd_schedule": wd_s, "momentum_schedule": mom_s,
         "teacher_temp_schedule": loss_mod.teacher_temp_schedule}
    reg_ids = {id(p) for p in groups[0]["params"]}
    names = [n for n, p in student.named_parameters()]
    g["param_names"] = np.array(names)
    g["regularized"] = np.array([id(p) in reg_ids for n, p in student.named_parameters()])
    g["requires_grad"] = np.array([p.requires_grad for n, p in student.named_parameters()])

    # the 96x96 positional table the student sees at its initial weights (interpolate_pos_encoding, vision_transformer.py:174-194)
    bb = student.backbone
    with torch.no_grad():
        x = torch.zeros(1, 37, 384)
        g["pos_embed_96"] = bb.interpolate_pos_encoding(x, 96, 96)[0].numpy().copy()
        g["cls_96"] = bb(synth.dino_crops(seed=300, B=2, n_local=1)[2]).numpy().copy()       # backbone CLS at 96 x 96

    for it in range(c["iters"]):
        epoch = it // c["niter_per_ep"]
        images = synth.dino_crops(seed=300 + it, B=c["B"], n_local=c["n_local"])
        for i, pg in enumerate(opt.param_groups):                  # main_dino.py:523-529
            pg["lr"] = lr_s[it]
            if i == 0:
                pg["weight_decay"] = wd_s[it]
        teacher_output = teacher(images[:2])
        student_output = student(images)
        center_before = loss_mod.center.clone()
        loss = loss_mod(student_output, teacher_output, epoch)
        opt.zero_grad()
        student_output.retain_grad()
        loss.backward()
        P = dict(student.named_parameters())
        if it == 0:
            for n in TRACK:
                if P[n].grad is not None:
                    g["grad0/" + n] = sample(P[n].grad)
            g["dlogits0"] = student_output.grad.numpy().copy()
        norms = utils.clip_gradients(student, c["clip_grad"])
        utils.cancel_gradients_last_layer(epoch, student, c["freeze_last_layer"])
        opt.step()
        with torch.no_grad():
            m = mom_s[it]
            for pq, pk in zip(student.parameters(), teacher.parameters()):
                pk.data.mul_(m).add_((1 - m) * pq.detach().data)
        k = f"it{it}/"
        g[k + "loss"] = np.array(loss.item())
        g[k + "teacher_out"] = teacher_output.detach().numpy().copy()
        g[k + "student_out"] = student_output.detach().numpy().copy()
        g[k + "center_before"] = center_before.numpy().copy()
        g[k + "center_after"] = loss_mod.center.numpy().copy()
        g[k + "norms"] = np.array(norms)
        g[k + "norm_names"] = np.array([n for n, p in student.named_parameters() if p.requires_grad])
        T = dict(teacher.named_parameters())
        for n in TRACK:
            g[k + "student/" + n] = sample(P[n])
            g[k + "teacher/" + n] = sample(T[n])
        print(f"it {it} epoch {epoch} loss {loss.item():.6f} lr {lr_s[it]:.3e} wd {wd_s[it]:.4f} m {m:.4f} "
              f"clipped {int((np.array(norms) > c['clip_grad']).sum())}/{len(norms)}")
    g["cfg_keys"] = np.array(sorted(CFG))
    g["cfg_vals"] = np.array([float(CFG[k]) for k in sorted(CFG)])
    np.savez_compressed(os.path.join(out, "dino_step.npz"), **g)


def golden_loss(main_dino, out):
    """DINOLoss alone at the reference's default out_dim = 65536, ncrops = 10 (2 + 8 local), non-zero centre."""
    n, B, ncrops = 65536, 2, 10
    gen = synth._gen(410)
    student_out = (torch.randn(ncrops * B, n, generator=gen) * 0.3).requires_grad_(True)
    teacher_out = torch.randn(2 * B, n, generator=gen) * 0.3
    mod = main_dino.DINOLoss(n, ncrops, 0.04, 0.07, 3, 10)
    mod.center = torch.randn(1, n, generator=gen) * 0.05
    c0 = mod.center.clone()
    g = {}
    for epoch in (0, 5):
        student_out.grad = None
        mod.center = c0.clone()
        loss = mod(student_out, teacher_out, epoch)
        loss.backward()
        g[f"e{epoch}/loss"] = np.array(loss.item())
        g[f"e{epoch}/grad_cols"] = student_out.grad[:, ::257].numpy().copy()
        g[f"e{epoch}/grad_rowsum"] = student_out.grad.double().sum(1).numpy()
        g[f"e{epoch}/grad_abs_sum"] = student_out.grad.double().abs().sum(1).numpy()
        g[f"e{epoch}/center_after"] = mod.center.numpy().copy()
        print("DINOLoss epoch", epoch, "loss", loss.item())
    np.savez_compressed(os.path.join(out, "dino_loss.npz"), **g)


def golden_args(main_dino, out):
    """Every flag of main_dino.get_args_parser() with its default and type: the CLI contract."""
    import json
    hub_list, torch.hub.list = torch.hub.list, lambda *a, **k: []      # main_dino.py:53 lists xcit archs over the network
    try:
        parser = main_dino.get_args_parser()
    finally:
        torch.hub.list = hub_list
    spec = {}
    for a in parser._actions:
        if not a.option_strings or a.dest == "help":
            continue
        d = a.default
        spec[a.option_strings[0]] = dict(dest=a.dest, default=list(d) if isinstance(d, tuple) else d,
                                         nargs=a.nargs, type=getattr(a.type, "__name__", None),
                                         store_true=type(a).__name__ == "_StoreTrueAction")
    with open(os.path.join(out, "dino_args.json"), "w") as f:
        json.dump(spec, f, indent=1, sort_keys=True)
    print("dino_args.json:", len(spec), "flags")


def main():
    torch.set_num_threads(8)
    main_dino, utils, vits = import_reference()
    golden_args(main_dino, HERE)
    init_dist()
    golden_loss(main_dino, HERE)
    golden_step(main_dino, utils, vits, HERE)


if __name__ == "__main__":
    main()
