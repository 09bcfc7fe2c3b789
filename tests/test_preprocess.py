"""Frame preprocessing: oracle vs Pillow (CPU), HIP kernel vs oracle and vs Pillow (GPU), bit for bit.

Pillow is the third-party library that owns the resampling arithmetic (Pillow==9.1.1 in the reference's
requirements.txt; 12.2.0 in this image — same 8-bit resampler); torchvision (absent) contributes only the crop box
formula and uint8/255 -> (x-mean)/std, restated in oracle/preprocess_oracle.py."""
import numpy as np
import pytest
import torch
from PIL import Image

from oracle import preprocess_oracle as po

SIZES = [(1080, 1920), (480, 854), (231, 517), (224, 224), (300, 200), (97, 1033), (720, 1280), (61, 45)]


def _frame(h, w, seed, smooth=False):
    g = np.random.default_rng(seed)
    if smooth:
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([127 + 120 * np.sin(xx / 37.0 + c) * np.cos(yy / 23.0 - c) for c in range(3)], -1)
        return np.clip(base + g.normal(0, 6, (h, w, 3)), 0, 255).astype(np.uint8)
    return g.integers(0, 256, (h, w, 3), dtype=np.uint8)


def _pil_pipeline(frame, hf=0.8, wf=0.8):
    """The reference's per-frame CPU pipeline with Pillow doing the work (torchvision 0.9.0 call sequence)."""
    img = Image.fromarray(frame)
    w, h = img.size
    ch, cw = hf * h, wf * w
    top, left = int(round((h - ch) / 2.)), int(round((w - cw) / 2.))
    img = img.crop((left, top, left + cw, top + ch)).resize((224, 224), Image.BILINEAR)
    t = torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).float().div(255)
    return t.sub_(torch.tensor(po.MEAN)[:, None, None]).div_(torch.tensor(po.STD)[:, None, None]).numpy()


@pytest.mark.parametrize("hw", SIZES)
def test_oracle_matches_pillow(hw):
    h, w = hw
    for seed, smooth in ((1, False), (2, True)):
        frame = _frame(h, w, seed, smooth)
        assert np.array_equal(po.preprocess_frame(frame), _pil_pipeline(frame))


def test_oracle_crop_box_and_lut():
    assert po.center_crop_box(1920, 1080) == (192, 108, 1728, 972)
    assert po.center_crop_box(854, 480) == (85, 48, 768, 432)
    assert po.center_crop_box(1280, 720, 0.8, 0.7) == (192, 72, 1088, 648)
    lut = po.normalize_lut()
    u8 = np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, 2)
    assert np.array_equal(lut[np.arange(3)[:, None, None], u8.transpose(2, 0, 1)], po.to_tensor_normalize(u8))


@pytest.mark.gpu
@pytest.mark.parametrize("hw", SIZES)
def test_hip_preprocess_bit_exact(hw):
    from sais_amd.preprocess import FramePreprocessor
    h, w = hw
    frames = np.stack([_frame(h, w, 10 + i, smooth=bool(i & 1)) for i in range(3)])
    pre = FramePreprocessor(h, w)
    assert pre.box == po.center_crop_box(w, h)
    got = pre(torch.from_numpy(frames).cuda()).cpu().numpy()
    for i in range(3):
        assert np.array_equal(got[i], po.preprocess_frame(frames[i])), f"frame {i} differs from the oracle"
        assert np.array_equal(got[i], _pil_pipeline(frames[i])), f"frame {i} differs from Pillow"
    pre.close()


@pytest.mark.gpu
def test_hip_preprocess_other_fractions_and_errors():
    from sais_amd import _lib as L
    from sais_amd.preprocess import FramePreprocessor
    h, w = 360, 640
    frames = np.stack([_frame(h, w, 77, True), _frame(h, w, 78)])
    pre = FramePreprocessor(h, w, 0.8, 0.7)
    got = pre(frames).cpu().numpy()                                  # numpy input is copied to the device
    for i in range(2):
        assert np.array_equal(got[i], _pil_pipeline(frames[i], 0.8, 0.7))
    with pytest.raises(ValueError):
        pre(torch.zeros(1, h, w + 1, 3, dtype=torch.uint8))
    with pytest.raises(ValueError):
        pre(torch.zeros(1, h, w, 3, dtype=torch.float32))
    with pytest.raises(L.SaisHipError):
        FramePreprocessor(h, w, device="cpu")
    with pytest.raises(L.SaisHipError):
        FramePreprocessor(h, w, 1.5, 0.8)
