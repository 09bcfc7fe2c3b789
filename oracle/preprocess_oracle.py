"""CPU ORACLE — TEST INFRASTRUCTURE ONLY (frame preprocessing).

numpy restatement of what the reference does to a decoded frame before the ViT
(SurgDataset.__getitem__, dino-main/main_dino.py:295-316, + the transform of
extract_representations.py:158-162):

    CenterCrop((0.8 H, 0.8 W)) -> Resize((224,224)) -> ToTensor -> Normalize(mean, std)

The arithmetic lives in two third-party dependencies that are absent from /root/reference:
  * torchvision==0.9.0 (requirements.txt): transforms.functional.center_crop / resize / to_tensor / normalize —
    restated from its published source: crop_top = int(round((H - ch) / 2.)), crop(left, top, left+cw, top+ch)
    with FLOAT ch, cw (the reference passes 0.8*H unrounded); PIL's Image.crop rounds each box edge with round();
    resize -> Image.resize(size[::-1], BILINEAR); to_tensor -> uint8.float().div(255); normalize -> sub_(mean).div_(std).
  * Pillow==9.1.1: Image.resize(BILINEAR) on 8-bit images = ImagingResample (src/libImaging/Resample.c):
    separable, antialiased (support = 1.0 * max(scale, 1)), coefficients normalised in double and rounded to
    22-bit fixed point, horizontal pass to uint8 first, then vertical pass; clip8((acc + 2^21) >> 22).

Parity status: PINNED against Pillow itself (12.2.0 is installed in this image, on the build container and on
the GPU box; the 8-bit resampler is unchanged since 9.1.1): tests/test_preprocess.py compares every function here
with PIL on random frames of awkward sizes, bit for bit.  torchvision is not installed: its four one-line formulas
above are restated from the published 0.9.0 source ("parity unpinned" for those lines only).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
OUT_SIZE = 224
MEAN = (0.485, 0.456, 0.406)            # extract_representations.py:148 (vit_small, SelfSupervised)
STD = (0.229, 0.224, 0.225)


def center_crop_box(width, height, height_frac=0.8, width_frac=0.8):
    """(left, top, right, bottom) integer box of CenterCrop((hf*H, wf*W)) on a PIL image."""
    ch, cw = height_frac * height, width_frac * width
    top = int(round((height - ch) / 2.))
    left = int(round((width - cw) / 2.))
    return tuple(int(round(v)) for v in (left, top, left + cw, top + ch))


def precompute_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter over the whole axis.
    Returns bounds int32 [out,2] (xmin, count) and integer coefficients int32 [out, ksize]."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [max(0.0, 1.0 - abs((x + xmin - center + 0.5) * ss)) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _clip8(acc):
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resample_axis0(img, out_size):
    """One pass of ImagingResample along axis 0 of a uint8 array [n, ...] -> [out_size, ...]."""
    bounds, kk = precompute_coeffs(img.shape[0], out_size)
    out = np.empty((out_size,) + img.shape[1:], np.uint8)
    for xx in range(out_size):
        xmin, cnt = bounds[xx]
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(cnt):
            acc += img[xmin + x].astype(np.int64) * int(kk[xx, x])
        out[xx] = _clip8(acc)
    return out


def resize_bilinear_u8(img, out_h=OUT_SIZE, out_w=OUT_SIZE):
    """Image.resize((out_w, out_h), BILINEAR) of a uint8 [H, W, C] array: horizontal pass first, then vertical."""
    tmp = resample_axis0(np.ascontiguousarray(img.transpose(1, 0, 2)), out_w).transpose(1, 0, 2)    # [H, out_w, C]
    return resample_axis0(np.ascontiguousarray(tmp), out_h)


def to_tensor_normalize(u8, mean=MEAN, std=STD):
    """ToTensor + Normalize: uint8 [H,W,C] -> float32 [C,H,W]; every step in float32 like torch."""
    x = u8.astype(np.float32) / np.float32(255)
    x = (x - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
    return np.ascontiguousarray(x.transpose(2, 0, 1))


def normalize_lut(mean=MEAN, std=STD):
    """[3,256] float32: the value ToTensor+Normalize gives each byte, per channel."""
    v = np.arange(256, dtype=np.float32)[None, :] / np.float32(255)
    return ((v - np.asarray(mean, np.float32)[:, None]) / np.asarray(std, np.float32)[:, None]).astype(np.float32)


def preprocess_frame(frame, height_frac=0.8, width_frac=0.8):
    """uint8 [H,W,3] decoded frame -> float32 [3,224,224] model input."""
    h, w = frame.shape[:2]
    l, t, r, b = center_crop_box(w, h, height_frac, width_frac)
    return to_tensor_normalize(resize_bilinear_u8(frame[t:b, l:r]))
