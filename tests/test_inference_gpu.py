"""BASELINE config 5 (long-video inference: 512 synthetic frames, hipGraph-captured ViT extraction, 34 sliding
windows x 3 TTA index sets, attention-map export) and the two-stream shapes of config 4, on a real MI355X,
checked against the CPU oracle; plus the main.sh-equivalent CLI run end to end on synthetic frames."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return True


def _models():
    from sais_amd.temporal import fullModel
    from sais_amd.vit import vit_small
    vit = vit_small(patch_size=16)
    vit.load_state_dict(synth.vit_state_dict(seed=0))
    m = fullModel('reps', 2, 'in_vs_out', 384, 'ViT', modalities='RGB-Flow')
    m.load_state_dict(synth.temporal_state_dict(seed=1))
    return vit.to(DEV).eval(), m.to(DEV).eval()


def test_long_video_512_frames_graph_extraction_and_windows(gpu):
    from oracle import sais_oracle as O
    from sais_amd.inference import FeatureExtractor, gesture_windows, run_windows, sample_window, tta_probs
    vit, m = _models()
    N = 512
    frames = synth.clips(seed=5, B=1, T=N)[0]                       # [512,3,224,224]
    flow_frames = synth.clips(seed=6, B=1, T=N // 15)[0]            # one flow map per 15 frames
    fx = FeatureExtractor(vit, batch_size=64, use_graph=True)
    reps = fx(frames.to(DEV))
    reps_eager = FeatureExtractor(vit, batch_size=64, use_graph=False)(frames.to(DEV))
    assert torch.equal(reps, reps_eager), "hipGraph replay must reproduce the eager kernels bit for bit"
    flow_reps = fx(flow_frames.to(DEV))
    # spot-check the features of 4 frames against the oracle (cosine against random directions <= 1e-3)
    idx = [0, 63, 64, 511]
    with torch.no_grad():
        ref = O.vit_forward(synth.vit_state_dict(seed=0), frames[idx])
    p = torch.randn(3, 384, generator=torch.Generator().manual_seed(5))
    cos = lambda a: (a / a.norm(dim=1, keepdim=True)) @ (p / p.norm(dim=1, keepdim=True)).t()
    assert (cos(reps[idx].cpu()) - cos(ref)).abs().max().item() <= 1e-3

    out, attention, _imp = run_windows(m, reps, flow_reps, videoname="synthetic", batch_size=2)
    wins = gesture_windows(N)
    assert len(wins) == 34 and len(out["reps"][0]) == 34 and len(out["videonames"]) == 34
    attn = torch.cat(attention)
    assert tuple(attn.shape) == (34, 16, 16)
    assert torch.allclose(attn.sum(-1), torch.ones(34, 16), atol=1e-4)
    # temporal half vs the oracle at the GPU's own features, incl. the wrap-around first window (index -1)
    tsd = synth.temporal_state_dict(seed=1)
    rc, fc = reps.cpu(), flow_reps.cpu()
    for w in (0, 1, 33):
        xs, fs = sample_window(rc, fc, *wins[w])
        assert [x.shape[1] for x in xs] == [15, 12, 9]
        for v in range(3):
            x, f = xs[v].unsqueeze(0), fs[v].unsqueeze(0)            # [1,1,T,384]
            padx = torch.zeros(1, 1, x.shape[2] + 1, dtype=torch.bool)
            padf = torch.zeros(1, 1, f.shape[2] + 1, dtype=torch.bool)
            with torch.no_grad():
                e_ref, a_ref = O.temporal_forward(tsd, x, f, padx, padf, "RGB-Flow")
            got = out["reps"][v][w]
            sim = O.cosine_logits(got.unsqueeze(0), synth.prototypes(2, 2))
            sim_ref = O.cosine_logits(e_ref, synth.prototypes(2, 2))
            assert (sim - sim_ref).abs().max().item() <= 1e-3
            if v == 0:
                assert (attn[w] - a_ref[0]).abs().max().item() <= 2e-3
    protos = torch.nn.ParameterDict({k: torch.nn.Parameter(v.to(DEV)) for k, v in synth.prototypes(2, 2).items()})
    probs = tta_probs(out, protos)
    assert tuple(probs.shape) == (34, 2) and torch.allclose(probs.sum(1).cpu(), torch.ones(34), atol=1e-5)


def test_tail_graph_and_batched_windows_equal_the_small_batch_forms(gpu):
    """Round 5: (a) a FeatureExtractor with a large main batch and a smaller captured tail shape gives, frame for frame, the
    features of the single-shape extractor (rows are independent; same kernels per dispatch regime up to the GEMM's tile
    walk: <= 1e-6 relative); (b) run_windows computing ALL windows per call equals the reference's batches of two run one
    by one (same outputs, same per-batch attention tensors) to fp32 summation order."""
    from sais_amd.inference import FeatureExtractor, run_windows
    vit, m = _models()
    frames = synth.clips(seed=9, B=1, T=150)[0].to(DEV)
    a = FeatureExtractor(vit, batch_size=128, use_graph=True, tail_batch=64)(frames)        # 128 + 64 (22 real frames)
    b = FeatureExtractor(vit, batch_size=64, use_graph=True)(frames)
    assert a.shape == (150, 384) and float((a - b).abs().max()) <= 2e-2 * float(b.abs().max())
    e = FeatureExtractor(vit, batch_size=128, use_graph=False, tail_batch=64)(frames)
    assert torch.equal(a, e), "graph replay of both captured shapes must reproduce the eager kernels bit for bit"
    # tail_batch="fit": the remainder gets a captured shape of its own size (150 = 128 + 22; the 9 "flow maps" another one)
    fit = FeatureExtractor(vit, batch_size=128, use_graph=True, tail_batch="fit")
    f = fit(frames)
    assert sorted(fit._graphs) == [22, 128] and float((f - b).abs().max()) <= 2e-2 * float(b.abs().max())
    assert torch.equal(f, FeatureExtractor(vit, batch_size=128, use_graph=False, tail_batch="fit")(frames))
    f9 = fit(frames[:9])                                                                     # 9 -> a shape of 10, one padded row
    assert sorted(fit._graphs) == [10, 22, 128] and torch.equal(f9, FeatureExtractor(vit, batch_size=128, use_graph=False,
                                                                                      tail_batch="fit")(frames[:9]))
    assert float((f9 - b[:9]).abs().max()) <= 2e-2 * float(b.abs().max())
    # at most MAX_GRAPHS captured shapes are kept: the least recently used remainder shape goes, never the main shape
    small = FeatureExtractor(vit, batch_size=8, use_graph=True, tail_batch="fit", tail_round=1)
    small.MAX_GRAPHS = 3
    for n in (8, 1, 2, 1, 3, 4):
        got = small(frames[:n])
        assert sorted(small._graphs)[-1] == 8 and len(small._graphs) <= 3
    assert sorted(small._graphs) == [3, 4, 8]
    assert torch.equal(got, FeatureExtractor(vit, batch_size=8, use_graph=False, tail_batch="fit", tail_round=1)(frames[:4]))
    reps = synth.reps(seed=3, B=1, T=200)[0, 0].to(DEV)
    flow = synth.reps(seed=4, B=1, T=13)[0, 0].to(DEV)
    big, attn_big, _ = run_windows(m, reps, flow, videoname="v", batch_size=2)                 # 13 windows in one call
    small, attn_small, _ = run_windows(m, reps, flow, videoname="v", batch_size=2, compute_batch=2)
    assert len(attn_big) == len(attn_small) == 7 and [tuple(x.shape) for x in attn_big] == [tuple(x.shape) for x in attn_small]
    for x, y in zip(attn_big, attn_small):
        assert float((x - y).abs().max()) <= 1e-5
    for v in range(3):
        assert len(big["reps"][v]) == 13
        for x, y in zip(big["reps"][v], small["reps"][v]):
            assert float((x - y).abs().max()) <= 1e-4 * max(1.0, float(y.abs().max()))
    # (c) the hipGraph form: fixed chunks of 32 windows (13 real + 19 repeats), flow rows padded to two under the mask; the
    # second video replays the SAME captured graph with new inputs (40 windows = two chunks)
    g1, attn_g1, _ = run_windows(m, reps, flow, videoname="v", batch_size=2, use_graph=True)
    assert len(getattr(m, "_window_graphs")) == 1 and len(attn_g1) == 7
    reps2 = synth.reps(seed=5, B=1, T=610)[0, 0].to(DEV)
    flow2 = synth.reps(seed=6, B=1, T=40)[0, 0].to(DEV)
    g2, attn_g2, _ = run_windows(m, reps2, flow2, videoname="w", batch_size=2, use_graph=True)
    e2, attn_e2, _ = run_windows(m, reps2, flow2, videoname="w", batch_size=2)
    assert len(getattr(m, "_window_graphs")) == 1 and len(g2["labels"]) == 40 == len(e2["labels"])
    for got, want, ag, aw in ((g1, big, attn_g1, attn_big), (g2, e2, attn_g2, attn_e2)):
        assert [tuple(x.shape) for x in ag] == [tuple(x.shape) for x in aw]
        for x, y in zip(ag, aw):
            assert float((x - y).abs().max()) <= 1e-5
        for v in range(3):
            for x, y in zip(got["reps"][v], want["reps"][v]):
                assert float((x - y).abs().max()) <= 1e-4 * max(1.0, float(y.abs().max()))


def test_two_streams_with_different_lengths_and_ragged_batch(gpu):
    """config 4 shapes at inference: RGB T=15/12, flow T=2/1 in one padded batch."""
    from oracle import sais_oracle as O
    _, m = _models()
    x = synth.reps(seed=21, B=2, T=15)
    f = synth.reps(seed=22, B=2, T=2)
    xl, fl = [15, 12], [2, 1]
    x[1, :, 12:] = 0
    f[1, :, 1:] = 0
    xpad, fpad = synth.padding_mask(xl), synth.padding_mask(fl)
    with torch.no_grad():
        emb, attn = m(x.to(DEV), f.to(DEV), xl, fl, 'Prototypes', xpad.to(DEV), fpad.to(DEV), None)
        e_ref, a_ref = O.temporal_forward(synth.temporal_state_dict(seed=1), x, f, xpad, fpad, "RGB-Flow")
    assert (emb.cpu() - e_ref).abs().max().item() <= 1e-4
    assert (attn.cpu() - a_ref).abs().max().item() <= 1e-5


def test_cli_main_sh_equivalent_on_synthetic_frames(gpu, tmp_path):
    """extract_representations.py (RGB + flow) -> run_experiments.py --inference, with the reference's flags, file
    layout and output formats."""
    from sais_amd import model_io
    from sais_amd.temporal import fullModel
    root = tmp_path / "SAIS"
    fold = root / "params" / "Fold_0"
    fold.mkdir(parents=True)
    m = fullModel('reps', 2, 'in_vs_out', 384, 'ViT')
    m.load_state_dict(synth.temporal_state_dict(seed=1))
    model_io.save_params_file(m, fold / "params.zip")
    model_io.save_prototypes_file(synth.prototypes(2, 2), fold / "prototypes.zip")
    env = dict(os.environ, PYTHONPATH=ROOT)
    sc = lambda name: os.path.join(ROOT, "SAIS/scripts", name)

    def main_sh(video, nframes):
        """the stages of SAIS/main.sh -f <video> -s <nframes> up to inference"""
        subprocess.run([sys.executable, sc("generate_paths.py"), "-f", video, "-p", str(root) + "/", "--synthetic_frames",
                        str(nframes)], check=True, env=env, cwd=ROOT)
        ex = [sys.executable, sc("extract_representations.py"), "--arch", "vit_small",
              "--patch_size", "16", "--model_type", "ViT_SelfSupervised_ImageNet", "--batch_size_per_gpu", "1024",
              "--data_path", str(root) + "/", "--data_list", "Custom", "--save_type", "h5", "--video", video,
              "--synthetic_frames", str(nframes)]
        subprocess.run(ex, check=True, env=env, cwd=ROOT)
        subprocess.run(ex + ["--optical_flow_to_reps"], check=True, env=env, cwd=ROOT)
        run = [sys.executable, sc("run_experiments.py"), "-p", str(root) + "/", "-data",
               "Custom_Gestures", "-d", "Custom", "-m", "ViT", "-enc", "ViT_SelfSupervised_ImageNet", "-t", "Prototypes",
               "-mod", "RGB-Flow", "-dim", "384", "-bs", "2", "-lr", "1e-1", "-nc", "2", "-bc", "-sa", "-domains",
               "in_vs_out", "-ph", "Custom_inference", "-dt", "reps", "-e", "1", "-f", "1", "--inference"]
        subprocess.run(run, check=True, env=env, cwd=ROOT)

    main_sh("vid_01", 64)
    r = torch.load(fold / "reps_and_labels_Custom_inference", weights_only=False)
    attn = torch.load(fold / "attention_Custom_inference", weights_only=False)
    imp = torch.load(fold / "importance_Custom_inference", weights_only=False)
    nwin = (64 - 15) // 15 + 1
    assert isinstance(r["reps"], tuple) and len(r["reps"]) == 3 and len(r["reps"][0]) == nwin
    assert tuple(r["reps"][0][0].shape) == (256,) and r["logits"] == [] and r["videonames"] == ["vid_01"] * nwin
    assert r["labels"][0].dtype == torch.int64 and r["labels"][0].dim() == 0
    assert sum(a.shape[0] for a in attn) == nwin and tuple(attn[0].shape[1:]) == (16, 16)
    assert imp == []
    # the features are a real HDF5 file with one dataset per video label (extract_representations.py:389-407)
    from sais_amd.hdf5_min import read_h5
    h5 = read_h5(str(root / "results" / "ViT_SelfSupervised_ImageNet_RepsAndLabels.h5"))
    assert list(h5.keys()) == ["vid_01"] and h5["vid_01"].shape == (64, 384) and h5["vid_01"].dtype.name == "float32"
    # a second, different video through the same project directory: the reps files are truncated (mode 'w') and the
    # windows come from the regenerated Custom_Paths.csv, so the outputs describe the new video only and the
    # post-processing stage lines up with them
    main_sh("vid_02", 48)
    r2 = torch.load(fold / "reps_and_labels_Custom_inference", weights_only=False)
    nwin2 = (48 - 15) // 15 + 1
    assert r2["videonames"] == ["vid_02"] * nwin2 and len(r2["reps"][0]) == nwin2
    assert list(read_h5(str(root / "results" / "ViT_SelfSupervised_ImageNet_RepsAndLabels.h5")).keys()) == ["vid_02"]
    # prototypes along +/- the mean embedding make every window a confident class-0 prediction (random prototypes give
    # p ~ 0.5: nothing survives the entropy gate and the reference's own script raises KeyError('Video') on the empty table)
    mean_emb = torch.stack(r2["reps"][0]).mean(0, keepdim=True)
    model_io.save_prototypes_file({"0": mean_emb, "1": -mean_emb}, fold / "prototypes.zip")
    subprocess.run([sys.executable, sc("process_inference_results.py"), "-p", str(root) + "/"], check=True, env=env, cwd=ROOT)
    lines = open(root / "results" / "Custom_inference_gestures.csv").read().strip().split("\n")
    assert len(lines) >= 2 and all(l.endswith("vid_02,images/vid_02") for l in lines[1:])


def test_main_sh_on_jpeg_frames_end_to_end(gpu, tmp_path):
    """bash SAIS/main.sh's stages on real JPEG files: generate_paths -> extract (decode on host, crop/resize/normalise
    on the GPU) -> flow reps -> inference -> post-processing CSV.  The RGB features must equal what the ViT gives on
    frames preprocessed by Pillow on the host (the reference's CPU pipeline)."""
    import numpy as np
    from PIL import Image
    from sais_amd import model_io
    from sais_amd.temporal import fullModel
    from SAIS.scripts._features_io import load_reps
    root = tmp_path / "SAIS"
    fold = root / "params" / "Fold_0"
    fold.mkdir(parents=True)
    m = fullModel('reps', 2, 'in_vs_out', 384, 'ViT')
    m.load_state_dict(synth.temporal_state_dict(seed=1))
    model_io.save_params_file(m, fold / "params.zip")
    model_io.save_prototypes_file(synth.prototypes(2, 2), fold / "prototypes.zip")
    nframes, h, w = 45, 180, 320
    g = np.random.default_rng(5)
    yy, xx = np.mgrid[0:h, 0:w]
    for sub, n in (("images", nframes), ("flows", nframes // 15)):
        (root / sub / "vid_01").mkdir(parents=True)
        for i in range(n):
            img = np.stack([127 + 100 * np.sin(xx / (11.0 + c) + i) * np.cos(yy / 17.0 - c) for c in range(3)], -1)
            img = np.clip(img + g.normal(0, 5, img.shape), 0, 255).astype(np.uint8)
            stem = "frames" if sub == "images" else "flows"
            Image.fromarray(img).save(root / sub / "vid_01" / f"{stem}_{i:08d}.jpg", quality=92)
    env = dict(os.environ, PYTHONPATH=ROOT)
    data = str(root) + "/"
    sc = lambda name: os.path.join(ROOT, "SAIS/scripts", name)
    subprocess.run([sys.executable, sc("generate_paths.py"), "-f", "vid_01", "-p", data], check=True, env=env, cwd=ROOT)
    ex = [sys.executable, sc("extract_representations.py"), "--arch", "vit_small", "--patch_size", "16", "--model_type",
          "ViT_SelfSupervised_ImageNet", "--batch_size_per_gpu", "1024", "--data_path", data, "--data_list", "Custom",
          "--save_type", "h5", "--video", "vid_01"]
    subprocess.run(ex, check=True, env=env, cwd=ROOT)
    subprocess.run(ex + ["--optical_flow_to_reps"], check=True, env=env, cwd=ROOT)
    subprocess.run([sys.executable, sc("run_experiments.py"), "-p", data, "-data", "Custom_Gestures", "-d", "Custom", "-m",
                    "ViT", "-enc", "ViT_SelfSupervised_ImageNet", "-t", "Prototypes", "-mod", "RGB-Flow", "-dim", "384",
                    "-bs", "2", "-lr", "1e-1", "-nc", "2", "-bc", "-sa", "-domains", "in_vs_out", "-ph",
                    "Custom_inference", "-dt", "reps", "-e", "1", "-f", "1", "--inference"], check=True, env=env, cwd=ROOT)
    # prototypes along +/- the mean embedding: every window becomes a confident class-0 prediction, so it survives the
    # entropy gate of the post-processing stage (random prototypes give p ~ 0.5 and an empty table)
    r = torch.load(fold / "reps_and_labels_Custom_inference", weights_only=False)
    mean_emb = torch.stack(r["reps"][0]).mean(0, keepdim=True)
    model_io.save_prototypes_file({"0": mean_emb, "1": -mean_emb}, fold / "prototypes.zip")
    subprocess.run([sys.executable, sc("process_inference_results.py"), "-p", data], check=True, env=env, cwd=ROOT)
    # features: same seeded ViT, frames preprocessed by Pillow on the host
    rgb = load_reps(data, "ViT_SelfSupervised_ImageNet_RepsAndLabels")["vid_01"]
    assert rgb.shape == (nframes, 384)
    from tests.test_preprocess import _pil_pipeline
    torch.manual_seed(0)
    vit = model_io.load_vit(None, device=torch.device("cuda:0"), drop_path_rate=0.1)
    files = sorted((root / "images" / "vid_01").glob("*.jpg"))
    host = np.stack([_pil_pipeline(np.asarray(Image.open(p))) for p in files])
    with torch.no_grad():
        ref = vit(torch.from_numpy(host).cuda()).cpu().numpy()
    np.testing.assert_allclose(rgb, ref, rtol=0, atol=1e-5)
    lines = open(root / "results" / "Custom_inference_gestures.csv").read().strip().split("\n")
    assert lines[0] == ",0,1,StartFrame,EndFrame,Entropy,pred,StartTime,EndTime,Gesture,Video,Path"
    assert len(lines) >= 2 and lines[1].endswith("vid_01,images/vid_01")


def test_cli_inference_sharded_over_two_ranks_equals_single_process(gpu, tmp_path):
    """SURVEY 8e, inference: frames (extract_representations.py) and window batches (run_experiments.py --inference) are
    sharded over the ranks of `torch.distributed.run`, results gathered in rank order, rank 0 writes.  Two ranks sharing
    this box's GPU (gloo) must write the same feature files, bit for bit, as one process, and the same window outputs up to fp32
    summation order."""
    from sais_amd import model_io
    from sais_amd.hdf5_min import read_h5
    from sais_amd.temporal import fullModel
    env = dict(os.environ, PYTHONPATH=ROOT, SAIS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    sc = lambda name: os.path.join(ROOT, "SAIS/scripts", name)
    outs = {}
    for world in (1, 2):
        root = tmp_path / f"w{world}" / "SAIS"
        fold = root / "params" / "Fold_0"
        fold.mkdir(parents=True)
        m = fullModel('reps', 2, 'in_vs_out', 384, 'ViT')
        m.load_state_dict(synth.temporal_state_dict(seed=1))
        model_io.save_params_file(m, fold / "params.zip")
        model_io.save_prototypes_file(synth.prototypes(2, 2), fold / "prototypes.zip")
        launch = [sys.executable] if world == 1 else \
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
             "127.0.0.1", "--master-port", str(29600 + os.getpid() % 300)]
        subprocess.run([sys.executable, sc("generate_paths.py"), "-f", "vid_01", "-p", str(root) + "/", "--synthetic_frames",
                        "100"], check=True, env=env, cwd=ROOT)
        ex = launch + [sc("extract_representations.py"), "--arch", "vit_small", "--patch_size", "16", "--model_type",
                       "ViT_SelfSupervised_ImageNet", "--batch_size_per_gpu", "64", "--data_path", str(root) + "/",
                       "--data_list", "Custom", "--save_type", "h5", "--video", "vid_01", "--synthetic_frames", "100",
                       "--tail_batch", "0"]                       # one padded shape: same kernels whatever the shard sizes
        subprocess.run(ex, check=True, env=env, cwd=ROOT)
        subprocess.run(ex + ["--optical_flow_to_reps"], check=True, env=env, cwd=ROOT)
        run = launch + [sc("run_experiments.py"), "-p", str(root) + "/", "-data", "Custom_Gestures", "-d", "Custom", "-m",
                        "ViT", "-enc", "ViT_SelfSupervised_ImageNet", "-t", "Prototypes", "-mod", "RGB-Flow", "-dim", "384",
                        "-bs", "2", "-lr", "1e-1", "-nc", "2", "-bc", "-sa", "-domains", "in_vs_out", "-ph",
                        "Custom_inference", "-dt", "reps", "-e", "1", "-f", "1", "--inference"]
        subprocess.run(run, check=True, env=env, cwd=ROOT)
        outs[world] = (read_h5(str(root / "results" / "ViT_SelfSupervised_ImageNet_RepsAndLabels.h5"))["vid_01"],
                       read_h5(str(root / "results" / "ViT_SelfSupervised_ImageNet_FlowRepsAndLabels.h5"))["vid_01"],
                       torch.load(fold / "reps_and_labels_Custom_inference", weights_only=False),
                       torch.load(fold / "attention_Custom_inference", weights_only=False))
    (rgb1, fl1, r1, a1), (rgb2, fl2, r2, a2) = outs[1], outs[2]
    assert rgb1.shape == (100, 384) and fl1.shape == (6, 384)
    assert (rgb1 == rgb2).all() and (fl1 == fl2).all()                  # fixed-shape graph batches: same kernels either way
    nwin = (100 - 15) // 15 + 1
    assert r2["videonames"] == r1["videonames"] == ["vid_01"] * nwin
    for v in range(3):
        # all TTA versions of a rank's windows run as ONE encoder pass (round 6): its row count, and with it the split-K summation
        # order of the N = 384 GEMMs, depends on how many windows the rank holds -> equal to fp32 rounding, no longer bit for bit
        assert len(r2["reps"][v]) == nwin and all(float((x - y).abs().max()) <= 5e-5 for x, y in zip(r1["reps"][v], r2["reps"][v]))
    # the head-averaged map is summed over the 4 heads with fp32 atomics: equal up to the order of those four additions
    assert len(a1) == len(a2) == (nwin + 1) // 2 and all(float((x - y).abs().max()) <= 1e-6 for x, y in zip(a1, a2))


def test_long_feature_file_runs_in_chunks_of_256_windows(gpu):
    """A 4 000-frame feature file = 266 windows: two calls of the model (256 + 10 windows, 4 096 token rows in the first) must give
    what batches of two give, window for window."""
    from sais_amd.inference import run_windows
    _, m = _models()
    reps = synth.reps(seed=31, B=1, T=4000)[0, 0].to(DEV)
    flow = synth.reps(seed=32, B=1, T=266)[0, 0].to(DEV)
    big, attn_big, _ = run_windows(m, reps, flow, videoname="long", batch_size=2)
    ref, attn_ref, _ = run_windows(m, reps, flow, videoname="long", batch_size=2, compute_batch=2)
    assert len(big["labels"]) == 266 == len(ref["labels"]) and len(attn_big) == 133 == len(attn_ref)
    for k in (0, 1, 63, 127, 128, 132):
        assert float((attn_big[k] - attn_ref[k]).abs().max()) <= 1e-5
    for v in range(3):
        for k in (0, 1, 2, 100, 255, 256, 265):
            x, y = big["reps"][v][k], ref["reps"][v][k]
            assert float((x - y).abs().max()) <= 1e-4 * max(1.0, float(y.abs().max()))


def test_bench_extract_workload_prints_one_valid_line(gpu):
    """`bench.py --workload extract` (BASELINE config 5 as a measured workload): one JSON line with the contract's fields, the in-run
    parity gate against the CPU oracle green, the two halves timed, a roofline object for the dominant kernel."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "extract", "--steps", "2", "--warmup", "1",
                        "--video-frames", "128", "--no-cpu-baseline"], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["unit"] == "frames/s" and d["value"] > 0 and d["n_gpus"] == 1 and d["scaling"] == "weak" and d["dtype"] == "bf16"
    assert d["config"]["video_frames"] == 128 and d["config"]["windows"] == (128 - 15) // 15 + 1 and "workload" in d["config"]
    assert d["parity"]["pass"] and d["parity"]["max_abs_logit"] <= 1e-3 and "invalid" not in d
    assert d["split_ms"]["vit_extraction"] > 0 and d["split_ms"]["windows_and_export"] > 0
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1 and d["roofline"]["all_kernels"]
