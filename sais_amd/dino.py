"""DINO self-distillation pre-training of the ViT-S/16 encoder on MI355X — drop-in for the training side of
SAIS/scripts/dino-main/main_dino.py (`train_one_epoch` :517-576, `DINOLoss` :579-630), `vision_transformer.DINOHead`
(:257-291) and the helpers of dino-main/utils.py it calls (`MultiCropWrapper` :595-630, `get_params_groups` :633-645,
`clip_gradients` :132-141, `cancel_gradients_last_layer` :144-149, `cosine_scheduler` :187-198), with the reference's
class / function names, argument meaning and checkpoint keys.

Every arithmetic step is a hand-written gfx950 kernel behind the C ABI (include/sais_hip.h): the student and teacher
ViTs are sais_amd.vit.VisionTransformer (197-token global crops, 37-token local crops), the head's Linear layers run on
the bf16x3 MFMA GEMM, and DINOLoss, the centre EMA, weight-norm, per-parameter clipping, AdamW and the EMA teacher are the
streaming kernels of csrc/dino.hip.  There is no autograd on this path (the step is forward kernels, backward kernels,
one optimizer pass per flat buffer) and no CPU fallback.

Differences from the reference, all deliberate: fp32 master weights with bf16 MFMA operands in the ViT instead of
torch.cuda.amp fp16 autocast + GradScaler (`--use_fp16`; no loss scaling is needed with bf16's exponent range);
`use_bn_in_head` is not supported (the reference's default is False); the PIL augmentation pipeline
(DataAugmentationDINO) is a CPU data-loader concern and stays outside this module: `images` is the list of collated
crops it returns, `[2 x [B,3,224,224]] + [n_local x [B,3,96,96]]`.
"""
import math

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib as L
from . import ops
from .flat import FlatParams
from .vit import VisionTransformer, _trunc_normal_, vit_small

F32 = torch.float32
FORCE_SYNC = False        # bench.py / tests: issue the data-parallel all-reduces even with a world of one


# --------------------------------------------------------------------------- schedules / groups (utils.py)
def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0):
    """utils.py:187-198 (same return type: a numpy array with one value per iteration)."""
    warmup_iters = warmup_epochs * niter_per_ep
    warmup = np.linspace(start_warmup_value, base_value, warmup_iters) if warmup_epochs > 0 else np.array([])
    iters = np.arange(epochs * niter_per_ep - warmup_iters)
    schedule = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * iters / len(iters)))
    return np.concatenate((warmup, schedule))


def get_params_groups(model):
    """utils.py:633-645: [{'params': regularized}, {'params': not_regularized, 'weight_decay': 0.}]."""
    regularized, not_regularized = [], []
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        (not_regularized if name.endswith(".bias") or len(param.shape) == 1 else regularized).append(param)
    return [{'params': regularized}, {'params': not_regularized, 'weight_decay': 0.}]


# --------------------------------------------------------------------------- DINOHead
class _WeightNormLinear(nn.Module):
    """nn.utils.weight_norm(nn.Linear(in, out, bias=False)): parameters weight_g [out,1], weight_v [out,in]
    (vision_transformer.py:277-281), registered in that order like torch's hook does."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        v = torch.empty(out_dim, in_dim)
        nn.init.kaiming_uniform_(v, a=math.sqrt(5))                  # nn.Linear's default init (the head's _init_weights
        self.weight_g = nn.Parameter(torch.ones(out_dim, 1))         # runs BEFORE last_layer exists, :275-277)
        self.weight_v = nn.Parameter(v)


class DINOHead(nn.Module):
    """vision_transformer.py:257-291.  forward(x f32 [R, in_dim]) -> logits f32 [R, out_dim]."""

    def __init__(self, in_dim, out_dim, use_bn=False, norm_last_layer=True, nlayers=3, hidden_dim=2048,
                 bottleneck_dim=256):
        super().__init__()
        if use_bn or nlayers != 3:
            raise NotImplementedError("DINOHead on MI355X: the reference's default head only (3 layers, no BatchNorm)")
        if in_dim % 128 or hidden_dim % 128 or bottleneck_dim % 128 or out_dim % 128 or bottleneck_dim > 1024:
            raise NotImplementedError("DINOHead on MI355X: layer widths must be multiples of 128 (GEMM tiles)")
        self.in_dim, self.hidden_dim, self.bottleneck_dim, self.out_dim = in_dim, hidden_dim, bottleneck_dim, out_dim
        self.mlp = nn.Sequential(nn.Linear(in_dim, hidden_dim), nn.GELU(), nn.Linear(hidden_dim, hidden_dim), nn.GELU(),
                                 nn.Linear(hidden_dim, bottleneck_dim))
        for m in self.mlp:
            if isinstance(m, nn.Linear):                             # _init_weights, :283-287
                _trunc_normal_(m.weight, std=.02)
                nn.init.constant_(m.bias, 0)
        self.last_layer = _WeightNormLinear(bottleneck_dim, out_dim)
        if norm_last_layer:
            self.last_layer.weight_g.requires_grad = False
        self.flat = None
        self._sig = None
        self._t_names = ["mlp.0.weight", "mlp.2.weight", "mlp.4.weight"]
        self._sentinels = ["mlp.0.weight", "mlp.4.bias", "last_layer.weight_v"]
        self._what = None                                            # normalised last-layer weight of the current params

    def _engine(self, device):
        if self.flat is None or not self.flat.intact() or self.flat.device != device:
            self.flat = FlatParams(self, device, f32_transposes=True)
            self._sig = None
        sig = self.flat.signature(self._sentinels)
        if sig != self._sig:
            self.flat.refresh_shadows(self._t_names)
            self._refresh_last_layer()
            self._sig = self.flat.signature(self._sentinels)
        return self.flat

    def shadows_dirty(self):
        self._sig = None

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._sig = None
        return r

    def _refresh_last_layer(self, need_backward=True):
        """w = g v / ||v||, its transpose (dX) and its bf16x3 image [hi | lo | hi] (the logits GEMM), once per parameter
        version."""
        f, dev = self.flat, self.flat.device
        if self._what is None or self._what[0].device != dev:
            self._what = (torch.empty(self.out_dim, self.bottleneck_dim, dtype=F32, device=dev),
                          torch.empty(self.bottleneck_dim, self.out_dim, dtype=F32, device=dev),
                          torch.empty(self.out_dim, dtype=F32, device=dev),
                          torch.empty(self.out_dim, 3 * self.bottleneck_dim, dtype=torch.bfloat16, device=dev))
        w, wt, inv, w3 = self._what
        ops.weight_norm_fwd(f.w32("last_layer.weight_v"), f.w32("last_layer.weight_g").view(-1), w, inv)
        if need_backward:
            ops.transpose_f32(w, self.out_dim, self.bottleneck_dim, wt)
        ops.split_bf16x3(w, w3, True)

    def after_optimizer_step(self, need_backward=True):
        if need_backward:
            self.flat._transposes(self._t_names)
        self.flat.epoch += 1
        self._refresh_last_layer(need_backward)
        self._sig = self.flat.signature(self._sentinels)

    def forward(self, x):
        return self.forward_kernels(x, save=False)[0]

    def forward_kernels(self, x, save):
        if not x.is_cuda:
            raise L.SaisHipError("DINOHead.forward needs a device tensor: the HIP path has no CPU fallback")
        f = self._engine(x.device)
        x = x.contiguous().float()
        R, dev = x.shape[0], x.device
        e = lambda *s: torch.empty(*s, dtype=F32, device=dev)
        u1, h1, u2, h2 = e(R, self.hidden_dim), e(R, self.hidden_dim), e(R, self.hidden_dim), e(R, self.hidden_dim)
        ops.gemm_nt_f32(x, f.w32("mlp.0.weight"), L.EPI_BIAS_F32, u1, bias=f.w32("mlp.0.bias"))
        ops.gelu_fwd_f32(u1, h1)
        ops.gemm_nt_f32(h1, f.w32("mlp.2.weight"), L.EPI_BIAS_F32, u2, bias=f.w32("mlp.2.bias"))
        ops.gelu_fwd_f32(u2, h2)
        z, zn, inv = e(R, self.bottleneck_dim), e(R, self.bottleneck_dim), e(R)
        ops.gemm_nt_f32(h2, f.w32("mlp.4.weight"), L.EPI_BIAS_F32, z, bias=f.w32("mlp.4.bias"))
        ops.l2norm_fwd(z, zn, inv)
        # 256 -> out_dim: bf16x3 as ONE bf16 GEMM over K' = 768 ([hi | hi | lo] . [hi | lo | hi]^T, fp32 accumulation)
        zn3 = torch.empty(R, 3 * self.bottleneck_dim, dtype=torch.bfloat16, device=dev)
        ops.split_bf16x3(zn, zn3, False)
        logits = e(R, self.out_dim)
        ops.gemm_nt(zn3, self._what[3], L.EPI_BIAS_F32, logits)
        saved = dict(x=x, u1=u1, h1=h1, u2=u2, h2=h2, zn=zn, inv=inv) if save else None
        return logits, saved

    def backward_kernels(self, saved, dlogits):
        """Accumulates the parameter gradients into self.flat.grad and returns dx f32 [R, in_dim]."""
        f = self.flat
        f.attach_grads()
        dev, R = dlogits.device, dlogits.shape[0]
        e = lambda *s: torch.empty(*s, dtype=F32, device=dev)
        w, wt, winv, _ = self._what
        dzn = e(R, self.bottleneck_dim)
        ops.gemm_nt_f32(dlogits, wt, L.EPI_BIAS_F32, dzn)
        dw = torch.zeros(self.out_dim, self.bottleneck_dim, dtype=F32, device=dev)
        ops.gemm_tn(dlogits, saved["zn"], dw)
        g = self.last_layer.weight_g
        ops.weight_norm_bwd(dw, f.w32("last_layer.weight_v"), f.w32("last_layer.weight_g").view(-1), winv,
                            f.g("last_layer.weight_v"), f.g("last_layer.weight_g").view(-1) if g.requires_grad else None)
        dz = e(R, self.bottleneck_dim)
        ops.l2norm_bwd(dzn, saved["zn"], saved["inv"], dz)
        ops.gemm_tn(dz, saved["h2"], f.g("mlp.4.weight"), f.g("mlp.4.bias"))
        dh2, du2 = e(R, self.hidden_dim), e(R, self.hidden_dim)
        ops.gemm_nt_f32(dz, f.wt16["mlp.4.weight"], L.EPI_BIAS_F32, dh2)
        ops.gelu_bwd_f32(dh2, saved["u2"], du2)
        ops.gemm_tn(du2, saved["h1"], f.g("mlp.2.weight"), f.g("mlp.2.bias"))
        dh1, du1 = dh2, e(R, self.hidden_dim)
        ops.gemm_nt_f32(du2, f.wt16["mlp.2.weight"], L.EPI_BIAS_F32, dh1)
        ops.gelu_bwd_f32(dh1, saved["u1"], du1)
        ops.gemm_tn(du1, saved["x"], f.g("mlp.0.weight"), f.g("mlp.0.bias"))
        dx = e(R, self.in_dim)
        ops.gemm_nt_f32(du1, f.wt16["mlp.0.weight"], L.EPI_BIAS_F32, dx)
        return dx


# --------------------------------------------------------------------------- MultiCropWrapper
class MultiCropWrapper(nn.Module):
    """utils.py:595-630: one backbone pass per run of equal-resolution crops, one head pass over all features."""

    def __init__(self, backbone, head):
        super().__init__()
        if not isinstance(backbone, VisionTransformer):
            raise NotImplementedError("MultiCropWrapper on MI355X wraps sais_amd.vit.VisionTransformer only")
        backbone.fc, backbone.head = nn.Identity(), nn.Identity()
        self.backbone = backbone
        self.head = head

    @staticmethod
    def _groups(x):
        if not isinstance(x, list):
            x = [x]
        groups, start = [], 0
        while start < len(x):
            end = start
            while end < len(x) and x[end].shape[-1] == x[start].shape[-1]:     # torch.unique_consecutive, :615-618
                end += 1
            groups.append(torch.cat(x[start:end]) if end - start > 1 else x[start])
            start = end
        return groups

    def forward(self, x):
        return self.forward_kernels(x, save=False)[0]

    def forward_kernels(self, x, save):
        """All resolution groups go through the backbone in ONE pass (rows stacked: sais_amd.vit._forward_kernels), the
        features come back in crop order, exactly the concatenation utils.py:626 builds."""
        bb = self.backbone
        groups = [bb._check_input(g) for g in self._groups(x)]
        bb._engine(groups[0].device)
        feat, bsaved = bb._forward_kernels(groups if len(groups) > 1 else groups[0], save=save)
        logits, hsaved = self.head.forward_kernels(feat, save)
        return logits, (bsaved, hsaved) if save else None

    def backward_kernels(self, saved, dlogits, sync=None):
        """sync (sais_amd.parallel.GradSync, data parallel): every slice of the two flat gradient buffers is handed to an
        asynchronous all-reduce the moment it is final — the head's right after the head backward, the backbone's block by
        block (last block first) — so that the exchange overlaps the remaining backward kernels; the caller joins with
        sync.wait() before the optimizer."""
        bsaved, hsaved = saved
        bb = self.backbone
        dfeat = self.head.backward_kernels(hsaved, dlogits)
        if sync is not None:
            sync._reduce(self.head.flat.grad)
            bb.grad_ready_hook = lambda a, b: sync._reduce(bb.flat.grad[a:b])
        try:
            bb._backward_kernels(bsaved, dfeat)
        finally:
            bb.grad_ready_hook = None


# --------------------------------------------------------------------------- DINOLoss
class DINOLoss(nn.Module):
    """main_dino.py:579-630.  forward(student_output, teacher_output, epoch) returns the loss (0-dim device tensor),
    leaves d loss / d student_output in `self.dlogits` (what loss.backward() would hand the student), and updates the
    centre — with the [1, out_dim] all-reduce of :627 when torch.distributed is initialised."""

    def __init__(self, out_dim, ncrops, warmup_teacher_temp, teacher_temp, warmup_teacher_temp_epochs, nepochs,
                 student_temp=0.1, center_momentum=0.9):
        super().__init__()
        self.student_temp = student_temp
        self.center_momentum = center_momentum
        self.ncrops = ncrops
        self.register_buffer("center", torch.zeros(1, out_dim))
        self.teacher_temp_schedule = np.concatenate((
            np.linspace(warmup_teacher_temp, teacher_temp, warmup_teacher_temp_epochs),
            np.ones(nepochs - warmup_teacher_temp_epochs) * teacher_temp))
        self.dlogits = None

    def forward(self, student_output, teacher_output, epoch):
        if not student_output.is_cuda:
            raise L.SaisHipError("DINOLoss.forward needs device tensors: the HIP path has no CPU fallback")
        dev = student_output.device
        if self.center.device != dev or self.center.dtype != F32:
            self.center = self.center.to(dev, F32)
        student_output, teacher_output = student_output.contiguous(), teacher_output.contiguous()
        B = teacher_output.shape[0] // 2
        temp = float(self.teacher_temp_schedule[epoch])
        center = self.center.view(-1)
        t_lse = ops.dino_row_lse(teacher_output, 1.0 / temp, center)
        s_lse = ops.dino_row_lse(student_output, 1.0 / self.student_temp)
        if self.dlogits is None or self.dlogits.shape != student_output.shape or self.dlogits.device != dev:
            self.dlogits = torch.empty_like(student_output)
        loss = torch.empty((), dtype=F32, device=dev)
        ops.dino_loss(student_output, teacher_output, center, s_lse, t_lse, B, self.ncrops, self.student_temp, temp,
                      self.dlogits, loss)
        self.update_center(teacher_output)
        return loss

    @torch.no_grad()
    def update_center(self, teacher_output):
        colsum = torch.empty(teacher_output.shape[1], dtype=F32, device=teacher_output.device)
        ops.dino_colsum(teacher_output, colsum)
        world = 1
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(colsum)
            world = dist.get_world_size()
        ops.dino_center_ema(self.center.view(-1), colsum, self.center_momentum, 1.0 / (teacher_output.shape[0] * world))


# --------------------------------------------------------------------------- optimizer tail
class _FlatAdamW:
    """AdamW state + chunk tables of ONE flat parameter buffer (the backbone's or the head's)."""

    def __init__(self, flat, teacher_flat, prefix, last_layer_names=()):
        self.flat, self.teacher_flat, self.prefix = flat, teacher_flat, prefix
        dev = flat.device
        self.exp_avg = torch.zeros_like(flat.flat)
        self.exp_avg_sq = torch.zeros_like(flat.flat)
        chunk = L.load().sais_opt_chunk_elems()
        rows, first, flags = [], [0], []
        for seg, (n, p) in enumerate(zip(flat.names, flat.params)):
            off, ln = flat.offsets[n], (p.numel() + 3) // 4 * 4
            fl = 0
            if not p.requires_grad:
                fl |= L.OPT_NO_GRAD
            elif not (n.endswith(".bias") or p.dim() == 1):          # get_params_groups: the regularized group
                fl |= L.OPT_DECAY
            if n in last_layer_names:
                fl |= L.OPT_CLASS1
            flags.append(fl)
            for o in range(0, ln, chunk):
                rows.append((off + o, min(chunk, ln - o), seg))
            first.append(len(rows))
        table = (L.SaisOptChunk * len(rows))(*[L.SaisOptChunk(*r) for r in rows])
        raw = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8)
        self.chunks = raw.to(dev)
        self.nchunks, self.nseg = len(rows), len(flags)
        self.seg_first = torch.tensor(first, dtype=torch.int32, device=dev)
        self.seg_flags = torch.tensor(flags, dtype=torch.int32, device=dev)
        self.partial = torch.empty(self.nchunks, dtype=F32, device=dev)
        self.norms = torch.zeros(self.nseg, dtype=F32, device=dev)
        self.flags = flags
        # segments that have a gradient, as a DEVICE index (a Python-list index would be a host -> device copy per call,
        # which a hipGraph capture cannot hold)
        self.keep_idx = torch.tensor([i for i, fl in enumerate(flags) if not fl & L.OPT_NO_GRAD], dtype=torch.int64, device=dev)

    def grad_norms(self, scale=1.0):
        ops.grad_norms(self.flat.grad, self.chunks, self.nchunks, self.seg_first, self.nseg, self.partial, self.norms, scale)
        return self.norms

    def step(self, clip, lr, wd, betas, eps, steps, frozen1, ema_m, with_shadow, grad_scale=1.0):
        a = L.SaisAdamW()
        f, t = self.flat, self.teacher_flat
        a.param, a.grad = f.flat.data_ptr(), f.grad.data_ptr()
        a.exp_avg, a.exp_avg_sq = self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr()
        a.teacher = t.flat.data_ptr() if t is not None else None
        a.param16 = f.w16.data_ptr() if with_shadow else None
        a.teacher16 = t.w16.data_ptr() if (with_shadow and t is not None) else None
        a.chunks, a.nchunks = self.chunks.data_ptr(), self.nchunks
        a.seg_flags, a.norms = self.seg_flags.data_ptr(), self.norms.data_ptr()
        a.clip, a.lr, a.weight_decay = float(clip or 0.0), float(lr), float(wd)
        a.beta1, a.beta2, a.eps = betas[0], betas[1], eps
        for c in range(2):
            s = max(steps[c], 1)
            a.bc1[c] = 1.0 - betas[0] ** s
            a.sqrt_bc2[c] = math.sqrt(1.0 - betas[1] ** s)
        a.frozen1, a.ema_m, a.grad_scale = int(frozen1), float(ema_m), float(grad_scale)
        ops.adamw_ema_step(a)


class DINOOptimizer:
    """Per-parameter clipping + AdamW on the two groups of get_params_groups + frozen last layer + EMA teacher
    (main_dino.py:541-566) as ONE kernel pass per flat buffer.  `param_groups` carries lr / weight_decay exactly like
    torch.optim.AdamW's so the schedule loop of train_one_epoch (:523-529) works unchanged."""

    def __init__(self, student, teacher, betas=(0.9, 0.999), eps=1e-8):
        self.student, self.teacher = student, teacher
        self.betas, self.eps = betas, eps
        groups = get_params_groups(student)
        self.param_groups = [dict(params=groups[0]['params'], lr=1e-3, weight_decay=1e-2, betas=betas, eps=eps),
                             dict(params=groups[1]['params'], lr=1e-3, weight_decay=0., betas=betas, eps=eps)]
        self.steps = [0, 0]                  # class 0: everything but the last layer; class 1: head.last_layer.*
        self._parts = None
        self.grad_scale = 1.0                # 1 / world when the flat gradients hold the all-reduced SUM over the ranks

    def _build(self):
        sb, sh = self.student.backbone, self.student.head
        tb, th = (self.teacher.backbone, self.teacher.head) if self.teacher is not None else (None, None)
        if sb.flat is None or sh.flat is None:
            raise L.SaisHipError("DINOOptimizer: run a forward pass (or call student.to(device) + _engine) first")
        if tb is not None and (tb.flat.names != sb.flat.names or th.flat.names != sh.flat.names):
            raise L.SaisHipError("DINOOptimizer: student and teacher must have the same parameter layout")
        self._parts = [(_FlatAdamW(sb.flat, tb.flat if tb is not None else None, "backbone."), True),
                       (_FlatAdamW(sh.flat, th.flat if th is not None else None, "head.",
                                   ("last_layer.weight_g", "last_layer.weight_v")), False)]

    def zero_grad(self, set_to_none=False):
        self._norms_fresh = False                        # new gradients: the norms of the last step no longer describe them
        for m in (self.student.backbone, self.student.head):
            if m.flat is not None:
                m.flat.grad.zero_()

    def clip_gradients(self):
        """utils.clip_gradients' return value (:132-141): the pre-clip L2 norm of every parameter that has a gradient,
        in named_parameters() order — as ONE device tensor (no host sync); the clipping itself happens in step()."""
        if self._parts is None:
            self._build()
        out = []
        for part, _ in self._parts:
            out.append(part.grad_norms(self.grad_scale).index_select(0, part.keep_idx))
        self._norms_fresh = True
        return torch.cat(out)

    def step(self, clip_grad=0.0, frozen_last_layer=False, ema_momentum=None):
        """One optimizer.step() + cancel_gradients_last_layer + teacher EMA.  clip_grad > 0 clips against the per-tensor
        norms of THESE gradients: clip_gradients() computes them (train_one_epoch's order); if it was not called since the
        last zero_grad() / step() they are computed here, so a stand-alone step() never clips against stale or zero norms."""
        if self._parts is None:
            self._build()
        if clip_grad and clip_grad > 0 and not getattr(self, "_norms_fresh", False):
            for part, _ in self._parts:
                part.grad_norms(self.grad_scale)
        self._norms_fresh = False
        lr, wd = self.param_groups[0]["lr"], self.param_groups[0]["weight_decay"]
        self.steps[0] += 1
        if not frozen_last_layer:
            self.steps[1] += 1
        for part, shadow in self._parts:
            part.step(clip_grad, lr, wd, self.betas, self.eps, self.steps, frozen_last_layer,
                      1.0 if ema_momentum is None else ema_momentum, shadow, self.grad_scale)
        for mod in (self.student, self.teacher):
            if mod is None:
                continue
            bb, trains = mod.backbone, mod is self.student
            if trains:                                   # the teacher only runs forward: no transposed shadows
                bb.flat._transposes(bb._t_names)
            bb.flat.epoch += 1
            bb._sig = bb.flat.signature(bb._sentinels)
            mod.head.after_optimizer_step(need_backward=trains)

    # torch.optim.AdamW-compatible checkpoint (main_dino.py:485-491: 'optimizer': optimizer.state_dict())
    def state_dict(self):
        index, state = {}, {}
        order = self.param_groups[0]["params"] + self.param_groups[1]["params"]
        for i, p in enumerate(order):
            index[id(p)] = i
        if self._parts is not None:
            for part, _ in self._parts:
                for n, p in zip(part.flat.names, part.flat.params):
                    if id(p) not in index:
                        continue
                    cls = 1 if part.prefix == "head." and n.startswith("last_layer.") else 0
                    if self.steps[cls] == 0:
                        continue
                    o = part.flat.offsets[n]
                    state[index[id(p)]] = dict(step=torch.tensor(float(self.steps[cls])),
                                               exp_avg=part.exp_avg[o:o + p.numel()].view(p.shape).cpu().clone(),
                                               exp_avg_sq=part.exp_avg_sq[o:o + p.numel()].view(p.shape).cpu().clone())
        n0 = len(self.param_groups[0]["params"])
        groups = []
        for gi, g in enumerate(self.param_groups):
            groups.append(dict(lr=g["lr"], betas=self.betas, eps=self.eps, weight_decay=g["weight_decay"], amsgrad=False,
                               maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                               params=list(range(0 if gi == 0 else n0, n0 if gi == 0 else n0 + len(g["params"])))))
        return dict(state=state, param_groups=groups)

    def load_state_dict(self, sd):
        if self._parts is None:
            self._build()
        order = self.param_groups[0]["params"] + self.param_groups[1]["params"]
        where = {}
        for part, _ in self._parts:
            for n, p in zip(part.flat.names, part.flat.params):
                where[id(p)] = (part, n)
        steps = [0, 0]
        for i, st in sd["state"].items():
            p = order[int(i)]
            part, n = where[id(p)]
            o = part.flat.offsets[n]
            part.exp_avg[o:o + p.numel()].copy_(st["exp_avg"].reshape(-1))
            part.exp_avg_sq[o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
            cls = 1 if part.prefix == "head." and n.startswith("last_layer.") else 0
            steps[cls] = max(steps[cls], int(float(st["step"])))
        self.steps = steps
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g["lr"], g["weight_decay"] = s["lr"], s["weight_decay"]


def clip_gradients(optimizer, clip):
    """utils.py:132-141 on the flat gradient buffers: returns the per-parameter norms; `clip` itself is applied inside
    optimizer.step(clip_grad=clip) (fused with AdamW — the clipped gradient is never written back)."""
    return optimizer.clip_gradients()


# --------------------------------------------------------------------------- the training step
def build_student_teacher(out_dim=65536, drop_path_rate=0.1, norm_last_layer=True, device="cuda:0", depth=12):
    """main_dino.py:375-421 for `--arch vit_small`: student / teacher MultiCropWrapper(ViT-S/16, DINOHead), teacher
    initialised from the student and frozen."""
    student = MultiCropWrapper(vit_small(patch_size=16, drop_path_rate=drop_path_rate, depth=depth),
                               DINOHead(384, out_dim, use_bn=False, norm_last_layer=norm_last_layer))
    teacher = MultiCropWrapper(vit_small(patch_size=16, depth=depth), DINOHead(384, out_dim, False))
    student, teacher = student.to(device), teacher.to(device)
    teacher.load_state_dict(student.state_dict())                    # :417
    for p in teacher.parameters():
        p.requires_grad = False
    return student, teacher


def _set_schedules(optimizer, it, lr_schedule, wd_schedule):
    for i, g in enumerate(optimizer.param_groups):                   # :523-529
        g["lr"] = float(lr_schedule[it])
        if i == 0:
            g["weight_decay"] = float(wd_schedule[it])


def _forward_backward(student, teacher, dino_loss, optimizer, images, epoch, clip_grad, want_norms):
    """main_dino.py:535-549: teacher / student forward, loss + centre update, zero_grad, backward (with the data-parallel
    gradient exchange), per-parameter norms.  No kernel argument in here changes from one iteration to the next inside an
    epoch (the teacher temperature is per EPOCH), which is what lets GraphedTrainStep capture it."""
    with torch.no_grad():
        groups = student._groups(images)         # one concatenation per resolution, shared by both networks
        two_global = groups[0] if groups[0].shape[0] == 2 * images[0].shape[0] else images[:2]
        teacher_output = teacher.forward_kernels(two_global, save=False)[0]          # :535  teacher(images[:2])
        student_output, saved = student.forward_kernels(groups, save=True)           # :536  student(images)
        loss = dino_loss(student_output, teacher_output, epoch)                      # :537
        optimizer.zero_grad()                                                        # :544
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        sync = None
        if world > 1 or (FORCE_SYNC and dist.is_initialized()):   # DistributedDataParallel's gradient averaging (:413): SUM here, 1 / world
            sync = getattr(optimizer, "_sync", None)                                 # inside the optimizer kernels
            if sync is None or sync.world != world:
                from .parallel import GradSync
                sync = optimizer._sync = GradSync(world, active=True)
        optimizer.grad_scale = 1.0 / world
        student.backward_kernels(saved, dino_loss.dlogits, sync)                     # :547
        if sync is not None:
            sync.wait()
        norms = None
        if clip_grad or want_norms:
            norms = clip_gradients(optimizer, clip_grad)                             # :548-549
    return loss, norms


def train_step(student, teacher, dino_loss, optimizer, images, it, epoch, lr_schedule, wd_schedule, momentum_schedule,
               clip_grad=3.0, freeze_last_layer=1, want_norms=False):
    """One iteration of train_one_epoch's loop body (main_dino.py:521-566).  Returns (loss 0-dim device tensor,
    per-parameter gradient norms or None).  Nothing here synchronises with the host."""
    _set_schedules(optimizer, it, lr_schedule, wd_schedule)
    loss, norms = _forward_backward(student, teacher, dino_loss, optimizer, images, epoch, clip_grad, want_norms)
    with torch.no_grad():
        optimizer.step(clip_grad=clip_grad or 0.0, frozen_last_layer=epoch < freeze_last_layer,      # :550-552
                       ema_momentum=float(momentum_schedule[it]))                    # :563-566
    return loss, norms


class GraphedTrainStep:
    """train_step with everything up to the optimizer replayed as ONE hipGraph: the two forward passes, the loss and centre
    update, zero_grad, the backward with its gradient exchange and the per-parameter norms (~350 launches) are captured; the
    optimizer tail (two fused clip + AdamW + EMA launches, the shadow refresh), whose lr / weight decay / momentum / bias
    corrections are kernel ARGUMENTS that change every iteration, stays eager.  The teacher temperature is a kernel argument
    too, but a per-EPOCH one: the graph is captured again when it changes (once per warm-up epoch).

    `images` are static device tensors: copy each new batch into them (`tensor.copy_`) before the call.  Capturing runs the
    step once for real (allocator, lazily built tables); the state that run touches — the centre, the DropPath RNG — is
    put back afterwards, so a graphed loop computes what the eager loop computes."""

    def __init__(self, student, teacher, dino_loss, optimizer, images, clip_grad=3.0, want_norms=False):
        self.student, self.teacher, self.dino_loss, self.optimizer = student, teacher, dino_loss, optimizer
        self.images, self.clip_grad, self.want_norms = list(images), clip_grad, want_norms
        self._graph, self._temp, self._out = None, None, None

    def step_state(self):
        """Snapshot of EVERYTHING the captured part of a step mutates besides the gradients (rebuilt every step): the DINO
        centre (EMA), the student's DropPath RNG state, the gradient exchange's byte / log counters.  One list, one place:
        whatever the captured region learns to mutate is added HERE, and test_graphed_train_step_equals_eager
        compares a graphed loop across a re-capture with the eager loop."""
        st = {"center": self.dino_loss.center.clone()}
        bb = self.student.backbone
        if bb._rng is not None:
            st["student_droppath_rng"] = bb._rng.clone()
        sync = getattr(self.optimizer, "_sync", None)
        if sync is not None:
            st["sync_counters"] = (sync.bytes, list(sync.log), list(sync.last_buckets))
        return st

    def load_step_state(self, st):
        with torch.no_grad():
            self.dino_loss.center.copy_(st["center"])
            if "student_droppath_rng" in st:
                self.student.backbone._rng.copy_(st["student_droppath_rng"])
        sync = getattr(self.optimizer, "_sync", None)
        if sync is not None and "sync_counters" in st:
            sync.bytes, sync.log, sync.last_buckets = st["sync_counters"][0], list(st["sync_counters"][1]), list(st["sync_counters"][2])

    def _capture(self, epoch):
        fn = lambda: _forward_backward(self.student, self.teacher, self.dino_loss, self.optimizer, self.images, epoch,
                                       self.clip_grad, self.want_norms)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()                                              # warm-up (its side effects are undone by the caller)
        torch.cuda.current_stream().wait_stream(side)
        self._graph = torch.cuda.CUDAGraph()
        # RCCL collectives inside (a gradient exchange is active): ProcessGroupNCCL's watchdog thread makes HIP calls of its own
        # while this thread captures.  A process group that merely exists (world 1, no exchange) keeps the strict mode.
        # An RCCL process group that exists is enough: DINOLoss.update_center all-reduces the column sums inside the
        # captured region whenever torch.distributed is initialised, world 1 included (main_dino.py always initialises one).
        import torch.distributed as dist
        sync = getattr(self.optimizer, "_sync", None)
        nccl = dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"
        mode = "thread_local" if (nccl or (sync is not None and getattr(sync, "active", False))) else "global"
        with torch.cuda.graph(self._graph, capture_error_mode=mode):
            self._out = fn()
        self._temp = float(self.dino_loss.teacher_temp_schedule[epoch])

    def __call__(self, it, epoch, lr_schedule, wd_schedule, momentum_schedule, freeze_last_layer=1):
        _set_schedules(self.optimizer, it, lr_schedule, wd_schedule)
        temp = float(self.dino_loss.teacher_temp_schedule[epoch])
        if self._graph is None or temp != self._temp:
            dev = self.images[0].device
            bb = self.student.backbone
            if self.dino_loss.center.device != dev:           # first call ever: DINOLoss would move it in the warm-up
                self.dino_loss.center = self.dino_loss.center.to(dev, F32)
            had_rng = bb._rng is not None
            pre = self.step_state()
            self._capture(epoch)
            if not had_rng and bb._rng is not None:           # the RNG state was created by the warm-up: back to its seed
                pre["student_droppath_rng"] = ops.rng_state(bb.drop_path_seed, dev)
            self.load_step_state(pre)                         # undo the warm-up run
        self._graph.replay()
        loss, norms = self._out
        self.optimizer._norms_fresh = norms is not None       # the replay computed them for THESE gradients
        with torch.no_grad():
            self.optimizer.step(clip_grad=self.clip_grad or 0.0, frozen_last_layer=epoch < freeze_last_layer,
                                ema_momentum=float(momentum_schedule[it]))
        return loss, norms


# --------------------------------------------------------------------------- checkpoints (main_dino.py:485-494)
def checkpoint_dict(student, teacher, optimizer, dino_loss, epoch, args=None):
    """The dict train_dino saves as checkpoint.pth: the student's keys carry DDP's `module.` prefix
    (`module.backbone.*`, `module.head.*` — what extract_representations.loadModel strips, :190-199), the teacher's
    do not (it is not wrapped when the head has no BatchNorm, :404-409)."""
    cpu = lambda sd, pre: {pre + k: v.detach().cpu().clone() for k, v in sd.items()}
    return {'student': cpu(student.state_dict(), 'module.'), 'teacher': cpu(teacher.state_dict(), ''),
            'optimizer': optimizer.state_dict(), 'epoch': epoch, 'args': args,
            'dino_loss': cpu(dino_loss.state_dict(), '')}


def load_checkpoint(ckpt, student, teacher, optimizer=None, dino_loss=None):
    """utils.restart_from_checkpoint (utils.py:152-184) for the objects above; returns the stored epoch."""
    student.load_state_dict({k[len('module.'):] if k.startswith('module.') else k: v for k, v in ckpt['student'].items()})
    teacher.load_state_dict({k[len('module.'):] if k.startswith('module.') else k: v for k, v in ckpt['teacher'].items()})
    if dino_loss is not None and 'dino_loss' in ckpt:
        dino_loss.load_state_dict(ckpt['dino_loss'])
    if optimizer is not None and 'optimizer' in ckpt:
        optimizer.load_state_dict(ckpt['optimizer'])
    return ckpt.get('epoch', 0)
