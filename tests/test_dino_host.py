"""CPU-only host logic of the DINO pre-training path (sais_amd/dino.py, sais_amd/vit.py): schedules, parameter groups,
state-dict names, the bicubic positional map.  No HIP compute."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import synth  # noqa: E402

from sais_amd import dino, vit  # noqa: E402

G = np.load(os.path.join(HERE, "golden", "dino_step.npz"))
CFG = {k: float(v) for k, v in zip(G["cfg_keys"], G["cfg_vals"])}


def test_pos_interp_matrix_is_torch_bicubic():
    """vision_transformer.py:174-194 uses F.interpolate(scale_factor=(n0 + 0.1) / 14, mode='bicubic')."""
    pos = torch.randn(1, 196, 8, dtype=torch.float64)
    for side in (96, 112, 160, 208):
        n0 = side // 16
        ref = F.interpolate(pos.reshape(1, 14, 14, 8).permute(0, 3, 1, 2), scale_factor=((n0 + 0.1) / 14, (n0 + 0.1) / 14),
                            mode="bicubic").permute(0, 2, 3, 1).reshape(-1, 8)
        M = vit.pos_interp_matrix(14, side, side)
        assert M.shape == (n0 * n0, 196)
        assert np.abs(M @ pos[0].numpy() - ref.numpy()).max() < 1e-12, side
    # the reference's own output at 96 x 96 (golden)
    p = synth.vit_state_dict(seed=20)["pos_embed"][0].double().numpy()
    got = np.concatenate([p[:1], vit.pos_interp_matrix(14, 96, 96) @ p[1:]])
    assert np.abs(got - G["pos_embed_96"]).max() < 2e-6


def test_schedules_match_reference():
    c = CFG
    lr = dino.cosine_scheduler(c["lr"] * c["B"] / 256.0, c["min_lr"], int(c["epochs"]), int(c["niter_per_ep"]),
                               warmup_epochs=int(c["warmup_epochs"]))
    wd = dino.cosine_scheduler(c["weight_decay"], c["weight_decay_end"], int(c["epochs"]), int(c["niter_per_ep"]))
    mom = dino.cosine_scheduler(c["momentum_teacher"], 1, int(c["epochs"]), int(c["niter_per_ep"]))
    np.testing.assert_allclose(lr, G["lr_schedule"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(wd, G["wd_schedule"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(mom, G["momentum_schedule"], rtol=0, atol=1e-15)
    loss = dino.DINOLoss(1024, 4, c["warmup_teacher_temp"], c["teacher_temp"], int(c["warmup_teacher_temp_epochs"]),
                         int(c["epochs"]))
    np.testing.assert_allclose(loss.teacher_temp_schedule, G["teacher_temp_schedule"], rtol=0, atol=1e-15)
    assert list(loss.state_dict()) == ["center"] and loss.center.shape == (1, 1024)


def test_parameter_contract_and_groups():
    student = dino.MultiCropWrapper(vit.vit_small(patch_size=16, drop_path_rate=0.1), dino.DINOHead(384, 1024))
    names = [n for n, _ in student.named_parameters()]
    assert names == list(G["param_names"])                       # the reference's named_parameters() order
    assert list(student.state_dict()) == names
    sd = {"backbone." + k: v for k, v in synth.vit_state_dict(seed=20).items()}
    sd.update({"head." + k: v for k, v in synth.dino_head_state_dict(seed=21, out_dim=1024).items()})
    student.load_state_dict(sd, strict=True)
    groups = dino.get_params_groups(student)
    reg = {id(p) for p in groups[0]["params"]}
    assert [id(p) in reg for _, p in student.named_parameters()] == [bool(x) for x in G["regularized"]]
    assert groups[1]["weight_decay"] == 0.0 and "weight_decay" not in groups[0]
    assert [p.requires_grad for _, p in student.named_parameters()] == [bool(x) for x in G["requires_grad"]]
    opt = dino.DINOOptimizer(student, None)
    osd = opt.state_dict()
    assert osd["state"] == {} and [len(g["params"]) for g in osd["param_groups"]] == [len(groups[0]["params"]),
                                                                                     len(groups[1]["params"])]
    # default construction follows the reference's initialisers
    head = dino.DINOHead(384, 1024, norm_last_layer=False)
    assert head.last_layer.weight_g.requires_grad and float(head.last_layer.weight_g.detach().min()) == 1.0
    assert float(head.mlp[0].bias.detach().abs().max()) == 0.0
    assert abs(float(head.mlp[2].weight.detach().std()) - 0.02) < 5e-4           # trunc_normal_(std=.02), cut at +-2
    assert float(head.last_layer.weight_v.detach().abs().max()) <= 1 / 16 + 1e-6  # kaiming_uniform(a=sqrt 5), fan_in 256


def test_checkpoint_keys_are_what_the_extraction_script_strips():
    """extract_representations.loadModel (:190-199): list(sd['student'].items())[:-8], name.split('.')[2:]."""
    student = dino.MultiCropWrapper(vit.vit_small(patch_size=16, depth=1), dino.DINOHead(384, 128))
    teacher = dino.MultiCropWrapper(vit.vit_small(patch_size=16, depth=1), dino.DINOHead(384, 128))
    ck = dino.checkpoint_dict(student, teacher, dino.DINOOptimizer(student, teacher), dino.DINOLoss(128, 4, .04, .04, 0, 2), 3)
    items = list(ck["student"].items())
    assert all(k.startswith("module.head.") for k, _ in items[-8:]) and all(k.startswith("module.backbone.") for k, _ in items[:-8])
    stripped = {".".join(k.split(".")[2:]): v for k, v in items[:-8]}
    assert list(stripped) == [k for k, _, _ in synth.vit_keys(depth=1)]
    assert ck["epoch"] == 3 and list(ck["dino_loss"]) == ["center"] and not any(k.startswith("module.") for k in ck["teacher"])


def test_no_cpu_fallback():
    import pytest
    from sais_amd._lib import SaisHipError
    with pytest.raises(SaisHipError):
        dino.DINOHead(384, 128)(torch.zeros(2, 384))
    with pytest.raises(SaisHipError):
        dino.DINOLoss(128, 4, .04, .04, 0, 2)(torch.zeros(8, 128), torch.zeros(4, 128), 0)


# ------------------------------------------------------------------ CLI + data pipeline
def _load_cli():
    import importlib.util
    path = os.path.join(os.path.dirname(HERE), "SAIS", "scripts", "dino-main", "main_dino.py")
    spec = importlib.util.spec_from_file_location("sais_main_dino", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_cli_flags_are_the_reference_parser():
    """tests/golden/dino_args.json = main_dino.get_args_parser() of the reference, dumped by make_golden_dino.py."""
    import json
    ref = json.load(open(os.path.join(HERE, "golden", "dino_args.json")))
    mod = _load_cli()
    mine = {a.option_strings[0]: a for a in mod.get_args_parser()._actions if a.option_strings}
    extra = set(mine) - set(ref)
    assert extra == {"--datasets", "--frames_root"} and not set(ref) - set(mine)
    for flag, spec in ref.items():
        a = mine[flag]
        d = list(a.default) if isinstance(a.default, tuple) else a.default
        assert a.dest == spec["dest"] and d == spec["default"], flag
        assert (type(a).__name__ == "_StoreTrueAction") == spec["store_true"], flag
        assert a.nargs == spec["nargs"], flag
        if spec["type"] is not None:
            assert getattr(a.type, "__name__", None) == spec["type"], flag
    ns = mod.get_args_parser().parse_args(["--norm_last_layer", "false", "--global_crops_scale", "0.14", "1"])
    assert ns.norm_last_layer is False and ns.global_crops_scale == [0.14, 1.0] and ns.use_fp16 is True


def test_augmentation_and_dataset(tmp_path):
    from PIL import Image
    import pandas as pd
    from sais_amd.dino_data import DataAugmentationDINO, SurgDataset, random_resized_crop
    import random
    rng = np.random.default_rng(0)
    root = tmp_path / "frames"
    (root / "Images" / "vidA").mkdir(parents=True)
    (tmp_path / "paths").mkdir()
    rows = []
    for i in range(5):
        arr = rng.integers(0, 256, (270, 480, 3), dtype=np.uint8)
        Image.fromarray(arr).save(root / "Images" / "vidA" / f"frames_{i:08d}.jpg", quality=95)
        rows.append((f"Images\\vidA\\frames_{i:08d}.jpg", "vidA"))                # Windows separators, as in the CSVs
    pd.DataFrame(rows, columns=["path", "label"]).to_csv(tmp_path / "paths" / "VUA_Paths.csv")
    aug = DataAugmentationDINO((0.4, 1.0), (0.05, 0.4), 3, seed=1)
    ds = SurgDataset(str(tmp_path), ["VUA"], aug, frames_root=str(root))
    assert len(ds) == 5
    crops, label, name = ds[2]
    assert [tuple(c.shape) for c in crops] == [(3, 224, 224)] * 2 + [(3, 96, 96)] * 3 and label == "vidA" and name == "VUA"
    assert all(c.dtype == torch.float32 and torch.isfinite(c).all() for c in crops)
    lo, hi = (0 - 0.485) / 0.229, (1 - 0.406) / 0.225
    assert min(float(c.min()) for c in crops) >= lo - 1e-4 and max(float(c.max()) for c in crops) <= hi + 1e-4
    # same seed -> same crops; the crop box statistics follow RandomResizedCrop's contract
    again = DataAugmentationDINO((0.4, 1.0), (0.05, 0.4), 3, seed=1)
    ds2 = SurgDataset(str(tmp_path), ["VUA"], again, frames_root=str(root))
    assert all(torch.equal(a, b) for a, b in zip(ds2[2][0], SurgDataset(str(tmp_path), ["VUA"], DataAugmentationDINO(
        (0.4, 1.0), (0.05, 0.4), 3, seed=1), frames_root=str(root))[2][0]))
    img = Image.fromarray(rng.integers(0, 256, (216, 384, 3), dtype=np.uint8))
    assert random_resized_crop(img, 96, (0.05, 0.4), random.Random(3)).size == (96, 96)
    assert ds.crop_fracs() == (0.8, 0.8)
    ds.dataset = "VUA_Gronau"
    assert ds.crop_fracs() == (0.8, 0.7)
    mod = _load_cli()
    crops_b, labels, names = mod.collate([ds[0], ds[1]])
    assert [tuple(c.shape) for c in crops_b] == [(2, 3, 224, 224)] * 2 + [(2, 3, 96, 96)] * 3 and len(labels) == 2


class _ConstImages(torch.utils.data.Dataset):
    """Every item is the SAME image: whatever differs between the returned crops comes from the augmentation draws."""

    def __init__(self, transform, n=8):
        from PIL import Image
        rng = np.random.default_rng(7)
        self.img = Image.fromarray(rng.integers(0, 256, (120, 160, 3), dtype=np.uint8))
        self.transform, self.n = transform, n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return self.transform(self.img)[0]                     # the first global crop


def test_augmentation_draws_differ_between_workers_and_between_epochs():
    """with one random.Random copied into every forked DataLoader worker, all workers drew the same
    crop sequence and every epoch replayed the first.  The generator is now keyed by (seed, worker id, torch's per-epoch
    worker seed)."""
    from sais_amd.dino_data import DataAugmentationDINO
    aug = DataAugmentationDINO((0.4, 1.0), (0.05, 0.4), 0, seed=3, global_size=32)
    loader = torch.utils.data.DataLoader(_ConstImages(aug), batch_size=2, num_workers=2, shuffle=False)
    torch.manual_seed(0)
    epoch0 = [b.clone() for b in loader]                       # batches 0, 2 come from worker 0; 1, 3 from worker 1
    epoch1 = [b.clone() for b in loader]
    assert len(epoch0) == 4
    assert not torch.equal(epoch0[0], epoch0[1])               # two workers, same items' image: different draws
    assert not torch.equal(epoch0[0][0], epoch0[0][1])         # consecutive draws of one worker differ
    assert all(not torch.equal(a, b) for a, b in zip(epoch0, epoch1))   # a new epoch does not replay the old one
    # the single-process path (num_workers = 0) is still reproducible from the seed
    a = DataAugmentationDINO((0.4, 1.0), (0.05, 0.4), 0, seed=3, global_size=32)
    b = DataAugmentationDINO((0.4, 1.0), (0.05, 0.4), 0, seed=3, global_size=32)
    ds = _ConstImages(None)
    assert torch.equal(a(ds.img)[0], b(ds.img)[0])
