"""Flat parameter / gradient / bf16-shadow storage shared by the ViT and temporal engines.

All parameters of a module live in ONE contiguous fp32 device buffer (the nn.Parameters are views
into it, so state_dict / load_state_dict / torch optimizers keep working), with a matching flat
fp32 gradient buffer (p.grad are views: the HIP backward kernels accumulate straight into it, the
data-parallel all-reduce runs on contiguous slices of it, and the fused SGD kernel updates the
whole model in one launch) and a flat bf16 shadow that the MFMA kernels read.
"""
import torch

from . import ops


class FlatParams:
    def __init__(self, module, device, f32_transposes=False):
        self.module = module
        self.f32_transposes = f32_transposes             # temporal engine: fp32 W^T for the bf16x3 GEMMs
        self.device = torch.device(device)
        named = [(n, p) for n, p in module.named_parameters()]
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        self.offsets = {}
        off = 0
        for n, p in named:
            self.offsets[n] = off
            off += (p.numel() + 3) // 4 * 4            # keep every tensor 16-B aligned
        self.numel = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.w16 = torch.zeros(off, dtype=torch.bfloat16, device=self.device)
        self.wt16 = {}                                   # name -> transposed bf16 copy [in, out]
        self.wt_buf = None                               # the buffer they are views of
        self._tr_key, self._tr_table = None, None         # descriptor table of the batched transpose
        with torch.no_grad():
            for n, p in named:
                o = self.offsets[n]
                self.flat[o:o + p.numel()].copy_(p.detach().reshape(-1).to(self.device, torch.float32))
                p.data = self.flat[o:o + p.numel()].view(p.shape)
        self._idx = {n: i for i, n in enumerate(self.names)}
        self.epoch = 0                                   # bumped by the fused SGD step
        self.attach_grads()

    # -- views ------------------------------------------------------------------------------
    def w32(self, name):
        i = self._idx[name]
        o = self.offsets[name]
        p = self.params[i]
        return self.flat[o:o + p.numel()].view(p.shape)

    def w(self, name):
        """bf16 shadow of a parameter, same shape."""
        o = self.offsets[name]
        p = self.params[self._idx[name]]
        return self.w16[o:o + p.numel()].view(p.shape)

    def g(self, name):
        o = self.offsets[name]
        p = self.params[self._idx[name]]
        return self.grad[o:o + p.numel()].view(p.shape)

    def intact(self):
        p0, p1 = self.params[0], self.params[-1]
        return (p0.data_ptr() == self.flat.data_ptr() + 4 * self.offsets[self.names[0]]
                and p1.data_ptr() == self.flat.data_ptr() + 4 * self.offsets[self.names[-1]])

    def attach_grads(self):
        """(Re)point every p.grad at its slice of the flat gradient buffer.  Returns True when the
        buffer had to be reset (e.g. after optimizer.zero_grad(set_to_none=True)).  Only a few
        sentinel parameters are inspected per call (the temporal model has 4118 tensors)."""
        base = self.grad.data_ptr()
        n_p = len(self.params)
        reset = False
        for i in {0, n_p // 3, (2 * n_p) // 3, n_p - 1}:
            p = self.params[i]
            if p.requires_grad and (p.grad is None or p.grad.data_ptr() != base + 4 * self.offsets[self.names[i]]):
                reset = True
        if reset:
            self.grad.zero_()
            for n, p in zip(self.names, self.params):
                if p.requires_grad:
                    o = self.offsets[n]
                    p.grad = self.grad[o:o + p.numel()].view(p.shape)
        return reset

    # -- bf16 shadows -----------------------------------------------------------------------
    def signature(self, sentinels):
        return (self.epoch,) + tuple(self.params[self._idx[n]]._version for n in sentinels)

    def _transposes(self, names):
        """Refresh every transposed shadow in one launch; the descriptor table lives on the device and is rebuilt only
        when the set of names changes (addresses are fixed: flat buffer + persistent shadow tensors)."""
        names = tuple(names)
        if not names:
            return
        if self._tr_key != names:
            entries = []
            # all transposed shadows live in ONE buffer (a single range to prefetch: ops.touch(self.wt_buf))
            dt = torch.float32 if self.f32_transposes else torch.bfloat16
            total = sum((self.params[self._idx[n]].numel() + 7) // 8 * 8 for n in names)
            self.wt_buf = torch.empty(total, device=self.device, dtype=dt)
            self.wt16, off = {}, 0
            for n in names:
                p = self.params[self._idx[n]]
                rows, cols = p.shape
                self.wt16[n] = self.wt_buf[off:off + rows * cols].view(cols, rows)
                off += (rows * cols + 7) // 8 * 8
                entries.append((self.w32(n), self.wt16[n]))
            self._tr_table = ops.transpose_table(entries, self.device)
            self._tr_key = names
        ops.transpose_batch(*self._tr_table, self.f32_transposes)

    def refresh_shadows(self, transposed_names):
        if not self.f32_transposes:
            ops.cast_bf16(self.flat, self.w16)
        self._transposes(transposed_names)

    def sgd_step(self, lr, grad_scale=1.0, transposed_names=()):
        ops.sgd_step(self.flat, self.grad, None if self.f32_transposes else self.w16, lr, grad_scale)
        self._transposes(transposed_names)
        self.epoch += 1
