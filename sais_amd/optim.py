"""Vanilla SGD (prepare_model.py:566-567: optim.SGD(params, lr); step at perform_training.py:155-158)
as one fused HIP launch per flat parameter buffer: w -= lr * g, bf16 MFMA shadow refreshed in the same
pass.  Parameters that do not belong to a sais_amd engine (the prototypes) get the same kernel each."""
import torch

from . import ops
from ._lib import SaisHipError


class SGD(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, engines=()):
        params = list(params)
        super().__init__(params, dict(lr=lr))
        self.engines = list(engines)          # modules exposing .flat / .sgd_step (VisionTransformer, fullModel)

    def _owned(self):
        owned = set()
        for e in self.engines:
            if e.flat is not None:
                owned.update(id(p) for p in e.flat.params)
        return owned

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        owned = self._owned()
        for group in self.param_groups:
            lr = group["lr"]
            for p in group["params"]:
                if id(p) in owned or p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()
                        and p.data_ptr() % 16 == 0 and p.grad.data_ptr() % 16 == 0):
                    raise SaisHipError("SGD.step: parameters must be contiguous, 16-B aligned fp32 device tensors "
                                       "(the HIP path has no torch / CPU fallback)")
                ops.sgd_step(p, p.grad, None, lr, grad_scale)
        lr = self.param_groups[0]["lr"]
        for e in self.engines:
            if e.flat is not None:
                e.sgd_step(lr, grad_scale)

    def zero_grad(self, set_to_none=False):
        for e in self.engines:
            if e.flat is not None:
                e.flat.grad.zero_()
        owned = self._owned()
        for group in self.param_groups:
            for p in group["params"]:
                if id(p) not in owned and p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.zero_()
