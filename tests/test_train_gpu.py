"""Training harness on a real MI355X: trainModel on synthetic feature files — phase loop, validation
early stopping, best snapshot, the files rank 0 writes — and the round trip of those files through
loadModel(inference=True), the reference's own consumer (prepare_model.py:517-570)."""
import csv
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _project(root, nvid=4, nframes=300, seed=0):
    """Feature files + annotated windows whose class is a planted direction in the RGB features."""
    from sais_amd.hdf5_min import write_h5
    rng = np.random.default_rng(seed)
    direction = rng.standard_normal(384).astype(np.float32)
    rgb, flow, rows = {}, {}, []
    for v in range(nvid):
        name = f"vid_{v:02d}"
        x = rng.standard_normal((nframes, 384)).astype(np.float32)
        for w, start in enumerate(range(0, nframes - 30, 30)):
            cls = (v + w) % 2
            x[start:start + 30] += (1.5 if cls else -1.5) * direction
            rows.append(dict(Video=name, Gesture=["in-view", "out-of-view"][cls], StartFrame=start + 1,
                             EndFrame=start + 31, phase="val" if v == nvid - 1 else "train"))
        rgb[name] = x
        flow[name] = rng.standard_normal((nframes // 15, 384)).astype(np.float32)
    os.makedirs(os.path.join(root, "results"))
    os.makedirs(os.path.join(root, "paths"))
    write_h5(os.path.join(root, "results", "ViT_SelfSupervised_ImageNet_RepsAndLabels.h5"), rgb)
    write_h5(os.path.join(root, "results", "ViT_SelfSupervised_ImageNet_FlowRepsAndLabels.h5"), flow)
    with open(os.path.join(root, "paths", "Custom_Gestures_Annotations.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=["Video", "Gesture", "StartFrame", "EndFrame", "phase"])
        w.writeheader()
        w.writerows(rows)
    return rows


def test_train_three_epochs_and_reload(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from sais_amd import train as T
    from sais_amd.model_io import loadModel
    root = str(tmp_path / "SAIS")
    rows = _project(root)
    savepath = os.path.join(root, "params", "Fold_0")
    torch.manual_seed(0)
    hist = T.trainModel(0, 1, root, savepath, "Custom_Gestures", "reps", 8, 2, "in_vs_out", ["train", "val"], 0.1,
                        "RGB-Flow", False, False, "Prototypes", True, False, False, "None", True, False, "ViT",
                        "ViT_SelfSupervised_ImageNet", 5, 1, 0, 384, 3, 0, 1)
    assert len(hist["loss"]) == 3 and hist["loss"][-1] < hist["loss"][0], hist["loss"]      # validation loss falls
    assert all(np.isfinite(v) for v in hist["loss"]) and 0.0 <= hist["acc"][-1] <= 1.0
    for f in ("params", "prototypes", "metrics", "reps_and_labels"):
        assert os.path.exists(os.path.join(savepath, f)), f
    raw = torch.load(os.path.join(savepath, "params"), weights_only=False)
    assert all(k.startswith("module.") for k in raw) and len(raw) == 4118
    r = torch.load(os.path.join(savepath, "reps_and_labels"), weights_only=False)
    nval = sum(1 for x in rows if x["phase"] == "val")
    assert isinstance(r["reps"], tuple) and len(r["reps"][0]) == nval and len(r["labels"]) == nval
    # README.md:64-75: rename to *.zip, then the reference's loader path
    shutil.copy(os.path.join(savepath, "params"), os.path.join(savepath, "params.zip"))
    shutil.copy(os.path.join(savepath, "prototypes"), os.path.join(savepath, "prototypes.zip"))
    md, _, dev = loadModel(0, 1, savepath, "reps", 2, "in_vs_out", 384, "ViT", "Prototypes", 0, lr=0.1,
                           modalities="RGB-Flow", inference=True)
    loaders, classes = T.load_dataloaders(root, "Custom_Gestures", 8, ["val"], "in_vs_out", "ViT_SelfSupervised_ImageNet")
    assert classes == ["in-view", "out-of-view"]
    md["model"].eval()                # what trainModel does for every phase but 'train' (train.py:57-63): no dropout
    metrics, snippets, labels, names, attn, imp, logits = T.single_epoch(0, 1, loaders, md, None, dev, "val", 2,
                                                                         "Prototypes", False)
    # the reloaded best snapshot reproduces the embeddings it was saved with (bit for bit: same kernels, same weights)
    for v in range(3):
        got = torch.stack([s.cpu() for s in snippets[v]])
        want = torch.stack(r["reps"][v])
        assert torch.equal(got, want), (v, (got - want).abs().max())
    assert abs(metrics["loss"] - min(hist["loss"])) < 1e-6 and tuple(attn[0].shape[1:]) == (11, 11)


def test_run_experiments_cli_trains(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = str(tmp_path / "SAIS")
    _project(root, nvid=3, nframes=150)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "SAIS/scripts/run_experiments.py"), "-p", root + "/", "-data",
                        "Custom_Gestures", "-d", "Custom", "-m", "ViT", "-enc", "ViT_SelfSupervised_ImageNet", "-t",
                        "Prototypes", "-mod", "RGB-Flow", "-dim", "384", "-bs", "4", "-lr", "1e-1", "-nc", "2", "-bc",
                        "-sa", "-domains", "in_vs_out", "-ph", "train", "val", "-dt", "reps", "-e", "2", "-f", "1"],
                       capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "All Info Saved!" in r.stdout and os.path.exists(os.path.join(root, "params", "Fold_0", "params"))


def _dp_train_worker(rank, world, port, root, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sais_amd import train as T
    # both ranks share cuda:0 here, so the device index trainModel derives from its `rank` argument is 0 for both; the data
    # shard is the true rank's (passed in as a ready dataloader) and each rank writes into its own directory
    loaders, _ = T.load_dataloaders(root, "Custom_Gestures", 4, ["train", "val"], "in_vs_out", "ViT_SelfSupervised_ImageNet",
                                    rank, world, seed=0)
    savepath = os.path.join(root, "params", f"Fold_0_r{rank}")
    torch.manual_seed(1234 + rank)          # every rank its own initial draws: trainModel must broadcast rank 0's
    hist = T.trainModel(0, world, root, savepath, "Custom_Gestures", "reps", 4, 2, "in_vs_out", ["train", "val"], 0.1,
                        "RGB-Flow", False, False, "Prototypes", True, False, False, "None", True, False, "ViT",
                        "ViT_SelfSupervised_ImageNet", 5, 1, 0, 384, 2, 0, 1, dataloader=loaders)
    torch.save(dict(hist=hist, ntrain=len(loaders["train"])), out + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_training_completes_with_replicas_in_step(tmp_path):
    """trainModel with world_size 2 (two ranks on this GPU, gloo): an ODD number of training windows (the shards are padded to
    equal length, every rank runs the same number of collective steps), averaged gradients — the job must finish and both
    ranks must see the same validation losses (= the same weights after every epoch)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    root = str(tmp_path / "SAIS")
    _project(root, nvid=4, nframes=330)
    ann = os.path.join(root, "paths", "Custom_Gestures_Annotations.csv")
    lines = open(ann).read().strip().split("\n")
    train_rows = [i for i, l in enumerate(lines) if l.endswith(",train")]
    if len(train_rows) % 2 == 0:
        del lines[train_rows[-1]]
    open(ann, "w").write("\n".join(lines) + "\n")
    assert sum(1 for l in lines if l.endswith(",train")) % 2 == 1
    out = str(tmp_path / "hist.pt")
    mp.spawn(_dp_train_worker, args=(2, 29800 + os.getpid() % 100, root, out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0", weights_only=False), torch.load(out + ".1", weights_only=False)
    assert r0["ntrain"] == r1["ntrain"]                          # same number of batches = same number of collectives
    h0, h1 = r0["hist"], r1["hist"]
    assert len(h0["loss"]) == 2 and all(np.isfinite(v) for v in h0["loss"])
    assert np.allclose(h0["loss"], h1["loss"], rtol=0, atol=1e-6), (h0["loss"], h1["loss"])
