"""TEST INFRASTRUCTURE — CPU oracle of the optical-flow stage, never imported by the product.

**PARITY UNPINNED.**  The reference computes its flow maps with a third-party model that is absent from /root/reference and
from this image: `ptlflow.get_model('raft', pretrained_ckpt='things')` (ptlflow==0.2.5; call sites
SAIS/scripts/extract_representations.py:33,62-67,221-252,267), whose pretrained 'things' checkpoint is unreachable offline.
This file restates RAFT *as published* — Teed & Deng, "RAFT: Recurrent All-Pairs Field Transforms for Optical Flow", ECCV
2020, and the authors' released implementation, which ptlflow's `raft` model ports layer for layer — in plain functional
fp32 torch, and `flow_to_rgb` as the Middlebury colour wheel in its flowpy form (what ptlflow.utils.flow_utils.flow_to_rgb
implements: bright background, radius normalised by the frame's own maximum).  No vector of the reference's exists for this
stage, so nothing here is checked against it: the tests hold the HIP kernels and the product module to THIS restatement.

Assumptions that cannot be verified offline (each is a parameter): 12 refinement iterations (the model's default `iters`),
inputs scaled to [-1, 1] from [0, 1] images in the channel order the frames are stored in, replicate padding to a multiple
of 8 split evenly between the two sides.

State dict keys follow the published model (fnet.*, cnet.*, update_block.*), so a `raft-things` checkpoint would load.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

HDIM = CDIM = 128
LEVELS, RADIUS = 4, 4


# --------------------------------------------------------------------------- encoders
def _norm(sd, pre, x, kind):
    if kind == "instance":                                   # nn.InstanceNorm2d: no affine, no running statistics
        return F.instance_norm(x, eps=1e-5)
    return F.batch_norm(x, sd[pre + "running_mean"], sd[pre + "running_var"], sd[pre + "weight"], sd[pre + "bias"],
                        training=False, eps=1e-5)


def _res_block(sd, pre, x, kind, stride):
    y = F.relu(_norm(sd, pre + "norm1.", F.conv2d(x, sd[pre + "conv1.weight"], sd[pre + "conv1.bias"], stride, 1), kind))
    y = F.relu(_norm(sd, pre + "norm2.", F.conv2d(y, sd[pre + "conv2.weight"], sd[pre + "conv2.bias"], 1, 1), kind))
    if stride != 1:
        x = _norm(sd, pre + "norm3.", F.conv2d(x, sd[pre + "downsample.0.weight"], sd[pre + "downsample.0.bias"], stride), kind)
    return F.relu(x + y)


def encoder(sd, pre, x, kind):
    """BasicEncoder: 7x7/2 conv, three stages of two residual blocks (64, 96/2, 128/2), 1x1 projection -> 1/8 resolution."""
    x = F.relu(_norm(sd, pre + "norm1.", F.conv2d(x, sd[pre + "conv1.weight"], sd[pre + "conv1.bias"], 2, 3), kind))
    for stage, stride in ((1, 1), (2, 2), (3, 2)):
        x = _res_block(sd, f"{pre}layer{stage}.0.", x, kind, stride)
        x = _res_block(sd, f"{pre}layer{stage}.1.", x, kind, 1)
    return F.conv2d(x, sd[pre + "conv2.weight"], sd[pre + "conv2.bias"])


# --------------------------------------------------------------------------- correlation volume
def corr_pyramid(f1, f2, levels=LEVELS):
    """All-pairs correlation <f1[:, :, i], f2[:, :, j]> / sqrt(C) as [B * H * W, 1, H, W], then 2 x 2 average pooling."""
    B, C, H, W = f1.shape
    corr = torch.matmul(f1.view(B, C, H * W).transpose(1, 2), f2.view(B, C, H * W)) / math.sqrt(C)
    pyr = [corr.reshape(B * H * W, 1, H, W)]
    for _ in range(levels - 1):
        pyr.append(F.avg_pool2d(pyr[-1], 2, stride=2))
    return pyr


def corr_lookup(pyr, coords, radius=RADIUS):
    """coords [B, 2, H, W] (x, y).  Per level l the 9 x 9 window around coords / 2^l, bilinear, zeros outside, pixel
    coordinates (grid_sample with align_corners=True).  Published quirk, kept: the window offsets are built as
    stack(meshgrid(dy, dx)) and added to (x, y), so channel 9 a + b of a level samples at (x + a - r, y + b - r)."""
    B, _, H, W = coords.shape
    c = coords.permute(0, 2, 3, 1)
    d = torch.linspace(-radius, radius, 2 * radius + 1)
    delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1).view(1, 2 * radius + 1, 2 * radius + 1, 2)
    out = []
    for l, corr in enumerate(pyr):
        Hl, Wl = corr.shape[-2:]
        pts = c.reshape(B * H * W, 1, 1, 2) / 2 ** l + delta
        gx = 2 * pts[..., 0] / (Wl - 1) - 1
        gy = 2 * pts[..., 1] / (Hl - 1) - 1
        s = F.grid_sample(corr, torch.stack([gx, gy], dim=-1), align_corners=True)
        out.append(s.view(B, H, W, -1))
    return torch.cat(out, dim=-1).permute(0, 3, 1, 2).contiguous()


# --------------------------------------------------------------------------- update operator
def _conv(sd, name, x, pad):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], padding=pad)


def motion_encoder(sd, flow, corr, pre="update_block.encoder."):
    cor = F.relu(_conv(sd, pre + "convc1", corr, 0))
    cor = F.relu(_conv(sd, pre + "convc2", cor, 1))
    flo = F.relu(_conv(sd, pre + "convf1", flow, 3))
    flo = F.relu(_conv(sd, pre + "convf2", flo, 1))
    out = F.relu(_conv(sd, pre + "conv", torch.cat([cor, flo], 1), 1))
    return torch.cat([out, flow], 1)


def sep_conv_gru(sd, h, x, pre="update_block.gru."):
    for k, pad in (("1", (0, 2)), ("2", (2, 0))):               # horizontal (1 x 5), then vertical (5 x 1)
        hx = torch.cat([h, x], 1)
        z = torch.sigmoid(_conv(sd, pre + "convz" + k, hx, pad))
        r = torch.sigmoid(_conv(sd, pre + "convr" + k, hx, pad))
        q = torch.tanh(_conv(sd, pre + "convq" + k, torch.cat([r * h, x], 1), pad))
        h = (1 - z) * h + z * q
    return h


def update_block(sd, net, inp, corr, flow):
    net = sep_conv_gru(sd, net, torch.cat([inp, motion_encoder(sd, flow, corr)], 1))
    dflow = _conv(sd, "update_block.flow_head.conv2", F.relu(_conv(sd, "update_block.flow_head.conv1", net, 1)), 1)
    mask = 0.25 * _conv(sd, "update_block.mask.2", F.relu(_conv(sd, "update_block.mask.0", net, 1)), 0)
    return net, mask, dflow


def upsample_flow(flow, mask):
    """[B, 2, H, W] -> [B, 2, 8 H, 8 W]: every fine pixel is a convex combination (softmax over mask) of the 3 x 3 coarse
    neighbours of 8 x flow."""
    B, _, H, W = flow.shape
    m = torch.softmax(mask.view(B, 1, 9, 8, 8, H, W), dim=2)
    up = F.unfold(8 * flow, [3, 3], padding=1).view(B, 2, 9, 1, 1, H, W)
    return torch.sum(m * up, dim=2).permute(0, 1, 4, 2, 5, 3).reshape(B, 2, 8 * H, 8 * W)


def pad_to_8(x):
    H, W = x.shape[-2:]
    ph, pw = (-H) % 8, (-W) % 8
    pads = [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]
    return F.pad(x, pads, mode="replicate"), pads


def raft_forward(sd, image1, image2, iters=12):
    """image1, image2: f32 [B, 3, H, W] in [0, 1].  Returns the flow image1 -> image2, f32 [B, 2, H, W]."""
    i1, pads = pad_to_8(2 * image1 - 1)
    i2, _ = pad_to_8(2 * image2 - 1)
    f1, f2 = encoder(sd, "fnet.", i1, "instance"), encoder(sd, "fnet.", i2, "instance")
    pyr = corr_pyramid(f1.float(), f2.float())
    c = encoder(sd, "cnet.", i1, "batch")
    net, inp = torch.tanh(c[:, :HDIM]), F.relu(c[:, HDIM:])
    B, _, H, W = f1.shape
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    coords0 = torch.stack([xs, ys]).unsqueeze(0).repeat(B, 1, 1, 1)
    coords1 = coords0.clone()
    up = None
    for _ in range(iters):
        corr = corr_lookup(pyr, coords1)
        net, mask, dflow = update_block(sd, net, inp, corr, coords1 - coords0)
        coords1 = coords1 + dflow
        up = upsample_flow(coords1 - coords0, mask)
    Hp, Wp = up.shape[-2:]
    return up[..., pads[2]:Hp - pads[3], pads[0]:Wp - pads[1]]


# --------------------------------------------------------------------------- colour coding
def colorwheel():
    """Middlebury wheel (Baker et al.): 55 hues, transitions RY 15, YG 6, GC 4, CB 11, BM 13, MR 6, linear in RGB."""
    hues = [(255, 0, 0), (255, 255, 0), (0, 255, 0), (0, 255, 255), (0, 0, 255), (255, 0, 255), (255, 0, 0)]
    rows = []
    for (a, b), n in zip(zip(hues[:-1], hues[1:]), (15, 6, 4, 11, 13, 6)):
        rows.append(np.linspace(np.array(a, dtype=np.float64), np.array(b, dtype=np.float64), n, endpoint=False))
    return np.concatenate(rows).astype(np.uint8).astype(np.float64)     # the published wheel is stored as uint8


def flow_to_rgb(flow, flow_max_radius=None):
    """flow f32 [..., 2, H, W] -> RGB f32 [..., 3, H, W] in [0, 1], bright background: hue from the flow angle (linear
    interpolation on the cyclic wheel), saturation from the radius normalised by `flow_max_radius` (default: the largest
    radius in `flow`); radii beyond 1 darken the hue instead."""
    f = flow.detach().cpu().double().numpy()
    u, v = f[..., 0, :, :], f[..., 1, :, :]
    radius, angle = np.hypot(u, v), np.arctan2(v, u)
    mx = radius.max() if flow_max_radius is None else flow_max_radius
    if mx > 0:
        radius = radius / mx
    wheel = colorwheel()
    n = len(wheel)
    angle = np.where(angle < 0, angle + 2 * np.pi, angle) * ((n - 1) / (2 * np.pi))
    wheel = np.vstack([wheel, wheel[:1]])
    frac, lo = np.modf(angle)
    hi = np.ceil(angle)
    hue = wheel[lo.astype(np.int64)] * (1 - frac[..., None]) + wheel[hi.astype(np.int64)] * frac[..., None]
    col = 255.0 - radius[..., None] * (255.0 - hue)
    oor = radius > 1
    col[oor] = hue[oor] / radius[oor][..., None]
    return torch.from_numpy(np.moveaxis(col.clip(0, 255) / 255.0, -1, -3).astype(np.float32))


def flow_image_uint8(rgb):
    """What the reference writes to flows_%08d.jpg: np.uint8(flow_rgb * 255) of the HWC image
    (extract_representations.py:245-249) — a truncation."""
    return np.uint8(rgb.permute(1, 2, 0).numpy() * 255)
