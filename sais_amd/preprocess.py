"""Frame preprocessing on the MI355X: decoded uint8 frames -> the float32 [F,3,224,224] ViT input.

Mirrors what the reference does on the CPU per frame (SurgDataset.__getitem__, dino-main/main_dino.py:295-316, and
the transform of extract_representations.py:158-162): CenterCrop((0.8 H, 0.8 W)) -> Resize((224,224)) -> ToTensor ->
Normalize.  The arithmetic (torchvision 0.9.0 crop box, Pillow's 8-bit antialiased bilinear resampler, float32
normalisation) is reproduced bit for bit by `sais_preprocess_run` (sais_amd/csrc/preprocess.hip); JPEG decoding stays
on the host.  The reference discards `img.convert('RGB')` (:297), so it effectively requires RGB frames; so does this.
"""
import ctypes

import numpy as np
import torch

from . import _lib as L

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)          # extract_representations.py:148 (vit_small, DINO)


def crop_fractions(dataset=None):
    """getCropDims, main_dino.py:318-323: (height_frac, width_frac)."""
    return (0.8, 0.7) if dataset in ('NS_Gronau', 'VUA_Gronau') else (0.8, 0.8)


class FramePreprocessor:
    """One plan per frame geometry (coefficient tables live on the device)."""

    def __init__(self, height, width, height_frac=0.8, width_frac=0.8, mean=MEAN, std=STD, device="cuda:0"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.SaisHipError("FramePreprocessor needs a GPU device: the HIP path has no CPU fallback")
        self.height, self.width = int(height), int(width)
        self._plan = ctypes.c_void_p()
        m = (ctypes.c_float * 3)(*mean)
        s = (ctypes.c_float * 3)(*std)
        with torch.cuda.device(self.device):
            L.call("sais_preprocess_plan_create", self.height, self.width, float(height_frac), float(width_frac),
                   ctypes.cast(m, ctypes.c_void_p), ctypes.cast(s, ctypes.c_void_p), ctypes.byref(self._plan))
        box = (ctypes.c_int * 4)()
        L.call("sais_preprocess_plan_box", self._plan, ctypes.cast(box, ctypes.c_void_p))
        self.box = tuple(box)                        # (left, top, right, bottom) of the centre crop

    def __call__(self, frames, out=None):
        """frames: uint8 [F, H, W, 3] on the device (or a numpy / CPU tensor, copied over) -> float32 [F,3,224,224]."""
        if isinstance(frames, np.ndarray):
            frames = torch.from_numpy(frames)
        if frames.dtype != torch.uint8 or frames.dim() != 4 or tuple(frames.shape[1:]) != (self.height, self.width, 3):
            raise ValueError(f"expected uint8 [F,{self.height},{self.width},3], got {frames.dtype} {tuple(frames.shape)}")
        frames = frames.to(self.device, non_blocking=True).contiguous()
        n = frames.shape[0]
        if out is None:
            out = torch.empty(n, 3, 224, 224, dtype=torch.float32, device=self.device)
        elif out.dtype != torch.float32 or tuple(out.shape) != (n, 3, 224, 224) or not out.is_contiguous() \
                or out.device != self.device:
            raise ValueError("out must be a contiguous float32 [F,3,224,224] tensor on the plan's device")
        L.call("sais_preprocess_run", self._plan, frames.data_ptr(), n, out.data_ptr(),
               torch.cuda.current_stream(self.device).cuda_stream)
        return out

    def close(self):
        if self._plan:
            L.load().sais_preprocess_plan_destroy(self._plan)
            self._plan = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
