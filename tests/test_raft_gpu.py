"""Optical-flow stage on a real MI355X (PARITY UNPINNED — see oracle/raft_oracle.py): the HIP correlation
volume (fp32-grade MFMA GEMM + one-pass pyramid pooling) and window lookup against the oracle, the product RAFT module end to
end against the oracle with the same seeded weights, the colour coding, and the `--optical_flow` stage of the CLI."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
import raft_synth  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return True


@pytest.mark.parametrize("H,W", [(16, 21), (28, 41), (68, 120)])
def test_correlation_pyramid_and_lookup_vs_oracle(gpu, H, W):
    from oracle import raft_oracle as R
    from sais_amd import ops
    g = torch.Generator().manual_seed(H * 100 + W)
    f1, f2 = torch.randn(1, 256, H, W, generator=g), torch.randn(1, 256, H, W, generator=g)
    ref = R.corr_pyramid(f1, f2)
    c0, lv = ops.raft_corr_pyramid(f1[0].to(DEV), f2[0].to(DEV))
    HW = H * W
    assert c0.shape[0] == HW and c0.stride(0) % 128 == 0
    scale = float(ref[0].abs().max())
    assert float((c0[:, :HW].cpu() - ref[0].view(HW, HW)).abs().max()) <= 2e-5 * scale       # bf16x3 split: fp32-grade
    for l in range(3):
        want = ref[l + 1].view(HW, -1)
        assert tuple(lv[l].shape) == tuple(want.shape)
        assert float((lv[l].cpu() - want).abs().max()) <= 2e-5 * scale
    # lookup: fractional coordinates, some far outside the image (zero padding) and on its border
    coords = torch.stack([torch.rand(H, W, generator=g) * (W + 12) - 6, torch.rand(H, W, generator=g) * (H + 12) - 6]).unsqueeze(0)
    coords[0, :, 0, 0] = torch.tensor([0.0, 0.0])
    coords[0, :, 1, 1] = torch.tensor([W - 1.0, H - 1.0])
    coords[0, :, 2, 2] = torch.tensor([-40.0, 3.0])
    got = ops.raft_lookup([(c0, lv)], coords.to(DEV).contiguous())
    want = R.corr_lookup(ref, coords)
    assert tuple(got.shape) == (1, 324, H, W)
    assert float((got.cpu() - want).abs().max()) <= 1e-4 * scale


@pytest.mark.parametrize("B,C,H,W,Cout,k,stride,pad,relu", [
    (2, 3, 50, 67, 64, 7, 2, 3, False),            # the encoders' stem
    (1, 64, 25, 34, 96, 3, 2, 1, False),           # a strided residual block
    (1, 96, 13, 17, 128, 1, 2, 0, False),          # its 1 x 1 downsample
    (2, 384, 16, 21, 128, (1, 5), 1, (0, 2), False),   # separable ConvGRU
    (1, 384, 16, 21, 128, (5, 1), 1, (2, 0), False),
    (1, 256, 16, 21, 126, 3, 1, 1, True),          # motion encoder (Cout not a multiple of anything) + fused ReLU
    (1, 256, 16, 21, 2, 3, 1, 1, False),           # flow head
    (1, 324, 16, 21, 256, 1, 1, 0, True)])
def test_conv2d_on_the_matrix_cores_vs_torch(gpu, B, C, H, W, Cout, k, stride, pad, relu):
    """ops.conv2d (im2col gather + bf16x3 GEMM, bias as a weight column) against torch.nn.functional.conv2d in fp64 on the
    host: every convolution shape RAFT uses."""
    import torch.nn.functional as F
    from sais_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + C + Cout)
    kh, kw = (k, k) if isinstance(k, int) else k
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(Cout, C, kh, kw, generator=g) / (C * kh * kw) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride, pad)
    if relu:
        ref = ref.relu()
    got = ops.conv2d(x.to(DEV), w.to(DEV), b.to(DEV), stride, pad, relu=relu)
    assert tuple(got.shape) == tuple(ref.shape)
    assert float((got.cpu().double() - ref).abs().max()) <= 3e-5 * max(1.0, float(ref.abs().max()))      # fp32-grade
    got2 = ops.conv2d(x.to(DEV), w.to(DEV), None, stride, pad, relu=relu)                              # no bias
    ref2 = F.conv2d(x.double(), w.double(), None, stride, pad)
    assert float((got2.cpu().double() - (ref2.relu() if relu else ref2)).abs().max()) <= 3e-5 * max(1.0, float(ref2.abs().max()))


def test_conv2d_weight_cache_follows_in_place_updates_and_freed_tensors(gpu):
    """ops.conv2d caches the padded [Cout, K + 1] weight matrix per live parameter tensor: an in-place update (a new
    checkpoint loaded into the module) and a different tensor at a recycled address must both rebuild it."""
    import torch.nn.functional as F
    from sais_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 8, 12, 15, generator=g).to(DEV)
    w, b = torch.randn(4, 8, 3, 3, generator=g).to(DEV), torch.randn(4, generator=g).to(DEV)
    y1 = ops.conv2d(x, w, b, 1, 1)
    w.mul_(2.0); b.add_(1.0)                                            # in place: same storage, new version
    y2 = ops.conv2d(x, w, b, 1, 1)
    ref2 = F.conv2d(x.double().cpu(), w.double().cpu(), b.double().cpu(), 1, 1)
    assert float((y2.cpu().double() - ref2).abs().max()) <= 5e-5 * float(ref2.abs().max()) and float((y2 - y1).abs().max()) > 0.1
    for seed in range(6):                                               # fresh tensors, old ones freed: addresses get reused
        w2 = torch.randn(4, 8, 3, 3, generator=torch.Generator().manual_seed(100 + seed)).to(DEV)
        y = ops.conv2d(x, w2, None, 1, 1)
        ref = F.conv2d(x.double().cpu(), w2.double().cpu(), None, 1, 1)
        assert float((y.cpu().double() - ref).abs().max()) <= 5e-5 * float(ref.abs().max())
        del w2, y


def test_raft_module_has_no_library_convolution(gpu):
    from sais_amd.raft import RAFT
    m = RAFT(iters=1)
    assert not any(isinstance(x, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)) for x in m.modules())
    assert sum(1 for x in m.modules() if type(x).__name__ == "_Conv") == 2 * 16 + 15     # two encoders + the update block


def test_raft_forward_and_colour_coding_vs_oracle(gpu):
    from oracle import raft_oracle as R
    from sais_amd.raft import RAFT, flow_image_uint8, flow_to_rgb
    sd = raft_synth.raft_state_dict(0)
    pairs = [raft_synth.frame_pair(s, 125, 164, shift=sh) for s, sh in ((1, (3, -2)), (2, (-1, 4)))]
    a, b = torch.cat([p[0] for p in pairs]), torch.cat([p[1] for p in pairs])
    m = RAFT(iters=4)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    flow = m(a.to(DEV), b.to(DEV))
    ref = R.raft_forward(sd, a, b, iters=4)
    assert tuple(flow.shape) == (2, 2, 125, 164)
    err = float((flow.cpu() - ref).abs().max())
    assert err <= 2e-2 * max(1.0, float(ref.abs().max())), err      # fp32 convolutions on two devices + the recurrence
    rgb = flow_to_rgb(flow[0])
    want = R.flow_to_rgb(flow[0].cpu())
    assert float((rgb.cpu() - want).abs().max()) <= 2e-3
    img = flow_image_uint8(rgb)
    assert img.shape == (125, 164, 3) and img.dtype == np.uint8
    assert int(np.abs(img.astype(np.int32) - R.flow_image_uint8(want).astype(np.int32)).max()) <= 1


def test_optical_flow_stage_writes_the_flow_images(gpu, tmp_path):
    """`extract_representations.py --optical_flow` (main.sh:18): one flows_<n:08d>.jpg per row of Custom_FlowPaths.csv,
    n = first frame // 15, the frame's own size, deterministic; a video whose flows folder exists is skipped."""
    from PIL import Image
    root = tmp_path / "SAIS"
    (root / "images" / "vid_01").mkdir(parents=True)
    g = np.random.default_rng(3)
    h, w = 136, 200
    yy, xx = np.mgrid[0:h, 0:w]
    for i in range(46):
        img = np.stack([127 + 100 * np.sin((xx + 0.4 * i) / (9.0 + c)) * np.cos((yy - 0.2 * i) / 13.0 - c) for c in range(3)], -1)
        Image.fromarray(np.clip(img + g.normal(0, 3, img.shape), 0, 255).astype(np.uint8)).save(
            root / "images" / "vid_01" / f"frames_{i:08d}.jpg", quality=92)
    env = dict(os.environ, PYTHONPATH=ROOT)
    data = str(root) + "/"
    sc = lambda name: os.path.join(ROOT, "SAIS/scripts", name)
    subprocess.run([sys.executable, sc("generate_paths.py"), "-f", "vid_01", "-p", data], check=True, env=env, cwd=ROOT)
    cmd = [sys.executable, sc("extract_representations.py"), "--arch", "vit_small", "--patch_size", "16", "--model_type",
           "ViT_SelfSupervised_ImageNet", "--batch_size_per_gpu", "2", "--data_path", data, "--data_list", "Custom",
           "--save_type", "h5", "--optical_flow", "--raft_iters", "3"]
    # no checkpoint and no explicit request for random weights: refuse (non-zero), write nothing
    r0 = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True)
    assert r0.returncode != 0 and "--raft_checkpoint" in (r0.stderr + r0.stdout) and not (root / "flows").exists()
    # a partial folder of an interrupted run does not count as done
    (root / "flows" / "vid_01").mkdir(parents=True)
    Image.fromarray(np.zeros((h, w, 3), np.uint8)).save(root / "flows" / "vid_01" / "flows_00000000.jpg")
    cmd = cmd + ["--raft_random_weights"]
    r = subprocess.run(cmd, check=True, env=env, cwd=ROOT, capture_output=True, text=True)
    assert "All Flows Saved!" in r.stdout and "3 flow maps saved" in r.stdout
    import json
    mk = json.load(open(root / "flows" / "vid_01" / ".flows_complete.json"))
    assert mk["count"] == 3 and mk["weights"].startswith("random")
    files = sorted(f for f in os.listdir(root / "flows" / "vid_01") if not f.startswith("."))
    assert files == ["flows_00000000.jpg", "flows_00000001.jpg", "flows_00000002.jpg"]       # frames 0, 15, 30 (+15 each)
    first = np.asarray(Image.open(root / "flows" / "vid_01" / files[0]))
    assert first.shape == (h, w, 3) and first.std() > 1.0
    r2 = subprocess.run(cmd, check=True, env=env, cwd=ROOT, capture_output=True, text=True)  # second run: nothing to do
    assert "0 flow maps saved" in r2.stdout and "already had flows" in r2.stdout
    # the flow images feed the next stage of main.sh
    subprocess.run([sys.executable, sc("extract_representations.py"), "--arch", "vit_small", "--patch_size", "16", "--model_type",
                    "ViT_SelfSupervised_ImageNet", "--batch_size_per_gpu", "256", "--data_path", data, "--data_list", "Custom",
                    "--save_type", "h5", "--optical_flow_to_reps", "--video", "vid_01"], check=True, env=env, cwd=ROOT)
    from SAIS.scripts._features_io import load_reps
    assert load_reps(data, "ViT_SelfSupervised_ImageNet_FlowRepsAndLabels")["vid_01"].shape == (3, 384)
