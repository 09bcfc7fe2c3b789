"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.

Functional restatement (state-dict in, tensors out; dtype follows the inputs, tests run it in fp64) of the DINO
pre-training objective the reference trains its ViT-S/16 encoder with.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product (sais_amd/dino.py) never does.

Parity status: PINNED.  tests/test_dino_oracle.py checks every function below against tests/golden/dino_step.npz /
dino_loss.npz, which tests/golden/make_golden_dino.py produced by running the reference's own main_dino.DINOLoss,
vision_transformer.DINOHead / VisionTransformer, utils.MultiCropWrapper / clip_gradients / cosine_scheduler /
get_params_groups and torch.optim.AdamW in the build container.

Paths are relative to /root/reference/SAIS/scripts/dino-main.  Gradients come from torch autograd on this restatement.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import sais_oracle as so


# --------------------------------------------------------------------------- positional table at other resolutions
def _cubic(x, A=-0.75):
    """Keys cubic-convolution kernel with A = -0.75: the coefficients of torch's upsample_bicubic2d
    (aten/src/ATen/native/UpSample.h: cubic_convolution1 / cubic_convolution2)."""
    x = abs(x)
    if x <= 1:
        return ((A + 2) * x - (A + 3)) * x * x + 1
    if x < 2:
        return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A
    return 0.0


def bicubic_matrix_1d(n_in, n_out, scale_factor):
    """[n_out, n_in] matrix of F.interpolate(mode='bicubic', align_corners=False, scale_factor=s) along one axis: source
    coordinate (dst + 0.5) / s - 0.5 (the given scale factor is used, not n_out / n_in), taps floor(x) - 1 .. + 2 with
    indices clamped to the border."""
    W = np.zeros((n_out, n_in), dtype=np.float64)
    for o in range(n_out):
        x = (o + 0.5) / scale_factor - 0.5
        x0 = math.floor(x)
        t = x - x0
        for k in range(-1, 3):
            idx = min(max(x0 + k, 0), n_in - 1)
            W[o, idx] += _cubic(k - t)
    return W


def pos_interp_matrix(n_side_in, w, h, patch=16):
    """[w0 * h0, n_side_in^2] linear map of interpolate_pos_encoding (vision_transformer.py:174-194): target grid
    (w // 16, h // 16), scale factors ((w0 + 0.1) / sqrt(N), (h0 + 0.1) / sqrt(N))."""
    w0, h0 = w // patch, h // patch
    sw, sh = (w0 + 0.1) / n_side_in, (h0 + 0.1) / n_side_in
    Ww = bicubic_matrix_1d(n_side_in, int(n_side_in * sw), sw)          # first interpolated axis (rows of the 14 x 14 grid)
    Wh = bicubic_matrix_1d(n_side_in, int(n_side_in * sh), sh)
    assert Ww.shape[0] == w0 and Wh.shape[0] == h0                      # :191
    return np.kron(Ww, Wh)


def interpolate_pos_encoding(pos_embed, npatch, w, h):
    """vision_transformer.py:174-194.  pos_embed [1, 197, D] -> [1, 1 + npatch, D]."""
    N = pos_embed.shape[1] - 1
    if npatch == N and w == h:
        return pos_embed
    side = int(math.sqrt(N))
    Wm = torch.as_tensor(pos_interp_matrix(side, w, h), dtype=pos_embed.dtype)
    patch_pos = Wm @ pos_embed[0, 1:]
    return torch.cat((pos_embed[:, 0], patch_pos), dim=0).unsqueeze(0)


def vit_forward_res(sd, x, depth=12, droppath=None):
    """VisionTransformer.forward at any multiple-of-16 resolution (prepare_tokens :196-207 + forward :209-214)."""
    tok = so.vit_patch_embed(sd, x)
    cls = sd["cls_token"].expand(tok.shape[0], -1, -1)
    t = torch.cat((cls, tok), dim=1)
    t = t + interpolate_pos_encoding(sd["pos_embed"], tok.shape[1], x.shape[2], x.shape[3])
    for i in range(depth):
        t = so.vit_block(sd, i, t, dp=None if droppath is None else (droppath[2 * i], droppath[2 * i + 1]))
    t = F.layer_norm(t, (t.shape[-1],), sd["norm.weight"], sd["norm.bias"], 1e-6)
    return t[:, 0]


# --------------------------------------------------------------------------- head, wrapper, loss
def dino_head(sd, x, pre="head."):
    """DINOHead.forward — vision_transformer.py:287-291: 3-layer MLP with exact-erf GELU, L2 normalisation
    (F.normalize eps 1e-12), weight-normalised bias-free last layer w = g v / ||v||_row (nn.utils.weight_norm, dim 0)."""
    h = F.gelu(F.linear(x, sd[pre + "mlp.0.weight"], sd[pre + "mlp.0.bias"]))
    h = F.gelu(F.linear(h, sd[pre + "mlp.2.weight"], sd[pre + "mlp.2.bias"]))
    z = F.linear(h, sd[pre + "mlp.4.weight"], sd[pre + "mlp.4.bias"])
    zn = z / z.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    v, g = sd[pre + "last_layer.weight_v"], sd[pre + "last_layer.weight_g"]
    w = g * v / v.norm(dim=1, keepdim=True)
    return zn @ w.t()


def multicrop_forward(sd, crops, depth=12, droppath=None):
    """MultiCropWrapper.forward — utils.py:611-630: one backbone pass per run of equal resolutions, features
    concatenated in crop order, one head pass.  sd keys: 'backbone.*', 'head.*'.  droppath: optional list with one
    [2 * depth, frames] factor table per resolution group."""
    bb = {k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}
    feats, start, gi = [], 0, 0
    while start < len(crops):
        end = start
        while end < len(crops) and crops[end].shape[-1] == crops[start].shape[-1]:
            end += 1
        feats.append(vit_forward_res(bb, torch.cat(crops[start:end]), depth,
                                     None if droppath is None else droppath[gi]))
        start, gi = end, gi + 1
    return dino_head(sd, torch.cat(feats))


def teacher_temp_schedule(warmup_teacher_temp, teacher_temp, warmup_epochs, nepochs):
    """DINOLoss.__init__ — main_dino.py:590-594."""
    return np.concatenate((np.linspace(warmup_teacher_temp, teacher_temp, warmup_epochs),
                           np.ones(nepochs - warmup_epochs) * teacher_temp))


def dino_loss(student_output, teacher_output, center, temp, ncrops, student_temp=0.1):
    """DINOLoss.forward — main_dino.py:596-619 (the centre update is `center_update`)."""
    s = (student_output / student_temp).chunk(ncrops)
    q = F.softmax((teacher_output - center) / temp, dim=-1).detach().chunk(2)
    total, n = 0, 0
    for iq in range(2):
        for v in range(ncrops):
            if v == iq:
                continue
            total = total + torch.sum(-q[iq] * F.log_softmax(s[v], dim=-1), dim=-1).mean()
            n += 1
    return total / n


def center_update(center, teacher_output, world_size=1, momentum=0.9, summed_over_ranks=None):
    """DINOLoss.update_center — main_dino.py:621-630.  summed_over_ranks: the all-reduced column sums when world > 1."""
    bc = teacher_output.sum(dim=0, keepdim=True) if summed_over_ranks is None else summed_over_ranks
    bc = bc / (len(teacher_output) * world_size)
    return center * momentum + bc * (1 - momentum)


# --------------------------------------------------------------------------- schedules, groups, optimizer
def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0):
    """utils.py:187-198."""
    warm = np.array([])
    wi = warmup_epochs * niter_per_ep
    if warmup_epochs > 0:
        warm = np.linspace(start_warmup_value, base_value, wi)
    it = np.arange(epochs * niter_per_ep - wi)
    sched = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * it / len(it)))
    return np.concatenate((warm, sched))


def is_regularized(name, shape):
    """get_params_groups — utils.py:633-645: biases and 1-D tensors are not weight-decayed."""
    return not (name.endswith(".bias") or len(shape) == 1)


def clip_coef(grad, clip):
    """clip_gradients — utils.py:132-141: PER-PARAMETER L2 clipping.  Returns (norm, factor applied)."""
    n = grad.norm(2)
    c = clip / (n + 1e-6)
    return n, (c if c < 1 else torch.ones_like(c))


def adamw_update(p, g, m, v, step, lr, wd, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.AdamW (single-tensor, amsgrad False, maximize False), as published in torch/optim/adamw.py /
    adam.py `_single_tensor_adam`: decoupled decay first, then the bias-corrected Adam step.  `step` is the 1-based
    count AFTER this update.  Returns (p, m, v)."""
    p = p * (1 - lr * wd)
    m = m + (g - m) * (1 - beta1)                        # exp_avg.lerp_(grad, 1 - beta1)
    v = v * beta2 + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v


def ema(teacher_p, student_p, m):
    """main_dino.py:563-566."""
    return teacher_p * m + (1 - m) * student_p


class TrainState:
    """Everything train_one_epoch mutates (main_dino.py:517-576), as plain dicts of tensors."""

    def __init__(self, student_sd, dtype=torch.float64):
        self.student = {k: v.detach().to(dtype).clone() for k, v in student_sd.items()}
        self.teacher = {k: v.clone() for k, v in self.student.items()}           # :417
        self.m = {k: torch.zeros_like(v) for k, v in self.student.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.student.items()}
        self.steps = {k: 0 for k in self.student}
        self.center = torch.zeros(1, student_sd["head.last_layer.weight_v"].shape[0], dtype=dtype)


def train_step(st, crops, it, epoch, lr_s, wd_s, mom_s, temp_s, clip_grad, freeze_last_layer, n_local, depth=12,
               droppath=None):
    """One iteration of train_one_epoch — main_dino.py:521-566 (fp32 branch).  Returns a dict with the loss, outputs,
    pre-clip gradient norms and the pre-clip gradients."""
    dt = st.center.dtype
    crops = [c.to(dt) for c in crops]
    trainable = [k for k in st.student if k != "head.last_layer.weight_g"]       # norm_last_layer=True, :280-281
    leaves = {k: st.student[k].clone().requires_grad_(k in trainable) for k in st.student}
    with torch.no_grad():
        teacher_out = multicrop_forward(st.teacher, crops[:2], depth)
    student_out = multicrop_forward(leaves, crops, depth, droppath)
    student_out.retain_grad()
    loss = dino_loss(student_out, teacher_out, st.center, float(temp_s[epoch]), n_local + 2)
    center_before = st.center.clone()
    st.center = center_update(st.center, teacher_out)
    loss.backward()
    grads = {k: leaves[k].grad.detach() for k in trainable}
    norms = {}
    lr, wd, mom = float(lr_s[it]), float(wd_s[it]), float(mom_s[it])
    for k in trainable:
        g = grads[k]
        n, c = clip_coef(g, clip_grad) if clip_grad else (g.norm(2), 1.0)
        norms[k] = float(n)
        if "last_layer" in k and epoch < freeze_last_layer:                      # utils.py:144-149: p.grad = None
            continue
        st.steps[k] += 1
        shape = st.student[k].shape
        st.student[k], st.m[k], st.v[k] = adamw_update(st.student[k], g * c, st.m[k], st.v[k], st.steps[k], lr,
                                                       wd if is_regularized(k, shape) else 0.0)
    for k in st.student:
        st.teacher[k] = ema(st.teacher[k], st.student[k], mom)
    return dict(loss=float(loss.detach()), teacher_out=teacher_out, student_out=student_out.detach(),
                dlogits=student_out.grad.detach(), norms=norms, grads=grads, center_before=center_before)
