"""Host-side data pipeline of DINO pre-training: the frame dataset and the multi-crop augmentation that
SAIS/scripts/dino-main/main_dino.py builds from torchvision.transforms (`SurgDataset` :266-326,
`DataAugmentationDINO` :633-679, utils.GaussianBlur / Solarization utils.py:36-68).

This is CPU data-loader work in front of the GPU step (sais_amd/dino.py) and outside the measured path; torchvision is
not a dependency of this repo, so the transforms are written directly on Pillow with the semantics torchvision documents
for its PIL backend (RandomResizedCrop's area / log-ratio sampling with 10 attempts and a centre-crop fallback,
ColorJitter's random order of brightness / contrast / saturation / hue, ...).  The random draws come from one
`random.Random` per worker, not from torch's global generator: augmentation parity is distributional, not bitwise
(DESIGN.md §7)."""
import math
import os
import random

import numpy as np
import torch
from PIL import Image, ImageEnhance, ImageFilter, ImageOps

MEAN = np.array((0.485, 0.456, 0.406), dtype=np.float32).reshape(3, 1, 1)
STD = np.array((0.229, 0.224, 0.225), dtype=np.float32).reshape(3, 1, 1)


def to_normalized_tensor(img):
    """ToTensor() + Normalize(ImageNet mean / std) (main_dino.py:643-646)."""
    a = np.asarray(img.convert("RGB"), dtype=np.float32).transpose(2, 0, 1) / 255.0
    return torch.from_numpy((a - MEAN) / STD)


def random_resized_crop(img, size, scale, rng, ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """transforms.RandomResizedCrop(size, scale=scale, interpolation=BICUBIC)."""
    W, H = img.size
    area = W * H
    box = None
    for _ in range(10):
        target = area * rng.uniform(scale[0], scale[1])
        ar = math.exp(rng.uniform(math.log(ratio[0]), math.log(ratio[1])))
        w, h = int(round(math.sqrt(target * ar))), int(round(math.sqrt(target / ar)))
        if 0 < w <= W and 0 < h <= H:
            top, left = rng.randint(0, H - h), rng.randint(0, W - w)
            box = (left, top, left + w, top + h)
            break
    if box is None:                                   # fallback: centre crop at the closest admissible ratio
        r = W / H
        if r < ratio[0]:
            w, h = W, int(round(W / ratio[0]))
        elif r > ratio[1]:
            w, h = int(round(H * ratio[1])), H
        else:
            w, h = W, H
        left, top = (W - w) // 2, (H - h) // 2
        box = (left, top, left + w, top + h)
    return img.crop(box).resize((size, size), Image.BICUBIC)


def color_jitter(img, rng, brightness=0.4, contrast=0.4, saturation=0.2, hue=0.1):
    """transforms.ColorJitter: the four adjustments in a random order, factors uniform in [1 - x, 1 + x] ([-hue, hue])."""
    ops = [0, 1, 2, 3]
    rng.shuffle(ops)
    fb, fc = rng.uniform(1 - brightness, 1 + brightness), rng.uniform(1 - contrast, 1 + contrast)
    fs, fh = rng.uniform(1 - saturation, 1 + saturation), rng.uniform(-hue, hue)
    for op in ops:
        if op == 0:
            img = ImageEnhance.Brightness(img).enhance(fb)
        elif op == 1:
            img = ImageEnhance.Contrast(img).enhance(fc)
        elif op == 2:
            img = ImageEnhance.Color(img).enhance(fs)
        else:
            h, s, v = img.convert("HSV").split()
            hh = (np.asarray(h, dtype=np.int16) + int(fh * 255)) % 256
            img = Image.merge("HSV", (Image.fromarray(hh.astype(np.uint8), "L"), s, v)).convert("RGB")
    return img


class DataAugmentationDINO:
    """main_dino.py:633-679: two global 224 x 224 views (blur p = 1.0 / blur p = 0.1 + solarize p = 0.2) and
    `local_crops_number` local 96 x 96 views (blur p = 0.5), each after flip / colour jitter (p = 0.8) / grayscale (p = 0.2)."""

    def __init__(self, global_crops_scale, local_crops_scale, local_crops_number, seed=None, global_size=224, local_size=96):
        self.gscale, self.lscale, self.local_crops_number = tuple(global_crops_scale), tuple(local_crops_scale), local_crops_number
        self.gsize, self.lsize = global_size, local_size
        self.seed = seed
        self._rng, self._rng_key = None, None

    @property
    def rng(self):
        """One random.Random per (process, DataLoader worker, epoch).  DataLoader workers are forked with a COPY of this
        object, re-forked every epoch from the parent's never-advancing state: a single generator made every worker draw
        the same crop / flip / jitter sequence and replay it every epoch.  torch gives each worker a seed
        `base_seed + worker_id` with a fresh base_seed per epoch (what torchvision's transforms draw from in the
        reference); the generator here is re-created from (seed, that worker seed) whenever it changes."""
        info = torch.utils.data.get_worker_info()
        key = (os.getpid(), None if info is None else (info.id, info.seed))
        if self._rng is None or key != self._rng_key:
            if info is None:
                self._rng = random.Random(self.seed)
            else:
                self._rng = random.Random(hash((self.seed, info.id, info.seed)) & 0xFFFFFFFFFFFF)
            self._rng_key = key
        return self._rng

    def _flip_and_color_jitter(self, img):
        r = self.rng
        if r.random() < 0.5:
            img = img.transpose(Image.FLIP_LEFT_RIGHT)
        if r.random() < 0.8:
            img = color_jitter(img, r)
        if r.random() < 0.2:
            img = img.convert("L").convert("RGB")
        return img

    def _blur(self, img, p):
        if self.rng.random() <= p:                                   # utils.GaussianBlur: radius ~ U(0.1, 2)
            img = img.filter(ImageFilter.GaussianBlur(radius=self.rng.uniform(0.1, 2.0)))
        return img

    def _view(self, image, size, scale, blur_p, solarize_p=0.0):
        img = random_resized_crop(image, size, scale, self.rng)
        img = self._blur(self._flip_and_color_jitter(img), blur_p)
        if solarize_p and self.rng.random() < solarize_p:            # utils.Solarization
            img = ImageOps.solarize(img)
        return to_normalized_tensor(img)

    def __call__(self, image):
        image = image.convert("RGB")
        crops = [self._view(image, self.gsize, self.gscale, 1.0), self._view(image, self.gsize, self.gscale, 0.1, 0.2)]
        crops += [self._view(image, self.lsize, self.lscale, 0.5) for _ in range(self.local_crops_number)]
        return crops


class SurgDataset(torch.utils.data.Dataset):
    """Frames listed in `<data_path>/paths/<dataset>_Paths.csv` (columns `path`, `label`; Windows separators allowed),
    border-cropped to the central 0.8 x 0.8 (0.8 x 0.7 for the *_Gronau sets) before the transform — main_dino.py:266-326
    with its NS / VUA branch (every row is a training row; DINO needs no labels).  Returns (crops, label, dataset)."""

    def __init__(self, data_path, dataset_list, transform, frames_root="./SAIS"):
        import pandas as pd
        rows = []
        for name in dataset_list:
            df = pd.read_csv(os.path.join(data_path, "paths", "%s_Paths.csv" % name), index_col=0)
            for p, lab in zip(df["path"].tolist(), df["label"].tolist() if "label" in df else [0] * len(df)):
                rows.append((str(p).replace("\\", "/"), lab, name))
        self.rows, self.transform, self.frames_root = rows, transform, frames_root
        self.dataset = dataset_list[0]

    def crop_fracs(self):
        return (0.8, 0.7) if self.dataset in ("NS_Gronau", "VUA_Gronau") else (0.8, 0.8)

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, idx):
        path, label, name = self.rows[idx]
        with open(os.path.join(self.frames_root, path), "rb") as f:
            img = Image.open(f)
            img.load()
        W, H = img.size
        hf, wf = self.crop_fracs()
        ch, cw = int(hf * H), int(wf * W)                            # transforms.CenterCrop((0.8 H, 0.8 W))
        left, top = int(round((W - cw) / 2.0)), int(round((H - ch) / 2.0))
        img = img.crop((left, top, left + cw, top + ch))
        return self.transform(img), label, name
