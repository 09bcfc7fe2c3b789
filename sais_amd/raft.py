"""Optical-flow stage of the SAIS pipeline on MI355X: RAFT flow between frames 15 apart, colour-coded and
written as the `flows/<video>/flows_%08d.jpg` images the flow stream of the ViT reads (reference:
SAIS/scripts/extract_representations.py:30-143,221-288 — `ptlflow.get_model('raft', pretrained_ckpt='things')`,
`flow_utils.flow_to_rgb`; main.sh:18).

PARITY UNPINNED: ptlflow 0.2.5 and its 'things' checkpoint are third-party artefacts absent from the reference tree and
unreachable offline.  The network follows the published RAFT (Teed & Deng, ECCV 2020) with the published parameter names, so a
`raft-things` state dict loads; without one the weights are seeded random (the stage then exercises the pipeline, not the
physics).  The arithmetic of the stage runs on this library's HIP kernels (include/sais_hip.h): the correlation volume — the
all-pairs GEMM (sais_gemm_nt_f32), its pyramid (sais_raft_corr_pool) and the per-iteration window lookup (sais_raft_lookup) —
and, since round 5, EVERY convolution of the encoders, the motion encoder, the separable ConvGRU, the flow head and the
upsampling-mask head: an im2col gather (sais_im2col_f32) + the fp32-grade bf16x3 matrix-core GEMM with the bias as one more
weight column (`ops.conv2d`; no MIOpen / nn.Conv2d on the path).  What stays in torch is element-wise glue: the instance /
batch normalisations (no learnable reduction across samples at inference), sigmoid / tanh of the GRU gates, concatenations,
the softmax of the 9-neighbour upsampling mask.  No CPU path: every op raises on host tensors.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

HDIM, LEVELS, RADIUS = 128, 4, 4


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class _Conv(nn.Module):
    """A convolution layer with nn.Conv2d's parameters (same names, shapes and default initialisation, so published state
    dicts load) whose forward is ops.conv2d: im2col + the bf16x3 matrix-core GEMM of this library.  relu=True fuses the
    activation that directly follows the layer into the GEMM's epilogue."""

    def __init__(self, cin, cout, kernel_size, stride=1, padding=0, relu=False):
        super().__init__()
        kh, kw = _pair(kernel_size)
        self.stride, self.padding, self.relu = _pair(stride), _pair(padding), relu
        self.weight = nn.Parameter(torch.empty(cout, cin, kh, kw))
        self.bias = nn.Parameter(torch.empty(cout))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))                  # nn.Conv2d.reset_parameters
        bound = 1.0 / math.sqrt(cin * kh * kw)
        nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        return ops.conv2d(x, self.weight, self.bias, self.stride, self.padding, relu=self.relu)


class _ResBlock(nn.Module):
    def __init__(self, cin, cout, norm, stride):
        super().__init__()
        self.conv1 = _Conv(cin, cout, 3, stride, 1)
        self.conv2 = _Conv(cout, cout, 3, 1, 1)
        self.norm1, self.norm2 = norm(cout), norm(cout)
        self.downsample = None
        if stride != 1:
            self.downsample = nn.Sequential(_Conv(cin, cout, 1, stride))
            self.norm3 = norm(cout)

    def forward(self, x):
        y = F.relu(self.norm1(self.conv1(x)))
        y = F.relu(self.norm2(self.conv2(y)))
        if self.downsample is not None:
            x = self.norm3(self.downsample(x))
        return F.relu(x + y)


class _Encoder(nn.Module):
    def __init__(self, out_dim, norm):
        super().__init__()
        self.conv1 = _Conv(3, 64, 7, 2, 3)
        self.norm1 = norm(64)
        self.layer1 = nn.Sequential(_ResBlock(64, 64, norm, 1), _ResBlock(64, 64, norm, 1))
        self.layer2 = nn.Sequential(_ResBlock(64, 96, norm, 2), _ResBlock(96, 96, norm, 1))
        self.layer3 = nn.Sequential(_ResBlock(96, 128, norm, 2), _ResBlock(128, 128, norm, 1))
        self.conv2 = _Conv(128, out_dim, 1)

    def forward(self, x):
        x = F.relu(self.norm1(self.conv1(x)))
        return self.conv2(self.layer3(self.layer2(self.layer1(x))))


class _MotionEncoder(nn.Module):
    def __init__(self, cor_planes):
        super().__init__()
        self.convc1, self.convc2 = _Conv(cor_planes, 256, 1, relu=True), _Conv(256, 192, 3, padding=1, relu=True)
        self.convf1, self.convf2 = _Conv(2, 128, 7, padding=3, relu=True), _Conv(128, 64, 3, padding=1, relu=True)
        self.conv = _Conv(256, 126, 3, padding=1, relu=True)

    def forward(self, flow, corr):                       # every ReLU of the block rides in its convolution's epilogue
        cor = self.convc2(self.convc1(corr))
        flo = self.convf2(self.convf1(flow))
        return torch.cat([self.conv(torch.cat([cor, flo], 1)), flow], 1)


class _SepConvGRU(nn.Module):
    def __init__(self, hidden=128, inp=256):
        super().__init__()
        for k, ks, pad in (("1", (1, 5), (0, 2)), ("2", (5, 1), (2, 0))):
            for n in "zrq":
                setattr(self, f"conv{n}{k}", _Conv(hidden + inp, hidden, ks, padding=pad))

    def forward(self, h, x):
        for k in "12":
            hx = torch.cat([h, x], 1)
            z = torch.sigmoid(getattr(self, "convz" + k)(hx))
            r = torch.sigmoid(getattr(self, "convr" + k)(hx))
            q = torch.tanh(getattr(self, "convq" + k)(torch.cat([r * h, x], 1)))
            h = (1 - z) * h + z * q
        return h


class _FlowHead(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1, self.conv2 = _Conv(128, 256, 3, padding=1, relu=True), _Conv(256, 2, 3, padding=1)

    def forward(self, x):
        return self.conv2(self.conv1(x))


class _UpdateBlock(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder = _MotionEncoder(LEVELS * (2 * RADIUS + 1) ** 2)
        self.gru = _SepConvGRU()
        self.flow_head = _FlowHead()
        self.mask = nn.Sequential(_Conv(128, 256, 3, padding=1), nn.ReLU(inplace=True), _Conv(256, 576, 1))     # keys mask.0 / mask.2

    def forward(self, net, inp, corr, flow):
        net = self.gru(net, torch.cat([inp, self.encoder(flow, corr)], 1))
        return net, 0.25 * self.mask(net), self.flow_head(net)


def _instance(c):
    return nn.InstanceNorm2d(c)


def _batch(c):
    return nn.BatchNorm2d(c)


class RAFT(nn.Module):
    """forward(image1, image2) -> flow f32 [B, 2, H, W] (image1 -> image2); images f32 [B, 3, H, W] in [0, 1] on the GPU."""

    def __init__(self, iters=12):
        super().__init__()
        self.iters = iters
        self.fnet = _Encoder(256, _instance)
        self.cnet = _Encoder(256, _batch)
        self.update_block = _UpdateBlock()

    def load_state_dict(self, sd, strict=True):
        # the published modules register norm3 twice (as `norm3` and as `downsample.1`): one copy is enough here
        sd = {k: v for k, v in sd.items() if ".downsample.1." not in k}
        return super().load_state_dict(sd, strict=strict)

    @staticmethod
    def _pad8(x):
        H, W = x.shape[-2:]
        ph, pw = (-H) % 8, (-W) % 8
        pads = [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]
        return F.pad(x, pads, mode="replicate"), pads

    @staticmethod
    def _upsample(flow, mask):
        B, _, H, W = flow.shape
        m = torch.softmax(mask.view(B, 1, 9, 8, 8, H, W), dim=2)
        nb = F.unfold(8 * flow, [3, 3], padding=1).view(B, 2, 9, 1, 1, H, W)
        return (m * nb).sum(2).permute(0, 1, 4, 2, 5, 3).reshape(B, 2, 8 * H, 8 * W)

    @torch.no_grad()
    def forward(self, image1, image2):
        if not image1.is_cuda:
            raise RuntimeError("sais_amd.raft.RAFT needs device tensors: the HIP path has no CPU fallback")
        i1, pads = self._pad8(2 * image1.float() - 1)
        i2, _ = self._pad8(2 * image2.float() - 1)
        f1, f2 = self.fnet(i1).float(), self.fnet(i2).float()
        B, _, H, W = f1.shape
        if min(H, W) < 16:
            raise ValueError("frames smaller than 128 x 128 leave the coarsest correlation level without a 2 x 2 grid")
        pyr = [ops.raft_corr_pyramid(f1[b], f2[b]) for b in range(B)]     # HIP: fp32-grade MFMA GEMM + one pooling pass
        c = self.cnet(i1)
        net, inp = torch.tanh(c[:, :HDIM]), F.relu(c[:, HDIM:])
        ys, xs = torch.meshgrid(torch.arange(H, device=f1.device, dtype=torch.float32),
                                torch.arange(W, device=f1.device, dtype=torch.float32), indexing="ij")
        coords0 = torch.stack([xs, ys]).unsqueeze(0).repeat(B, 1, 1, 1).contiguous()
        coords1 = coords0.clone()
        up = None
        for _ in range(self.iters):
            corr = ops.raft_lookup(pyr, coords1.contiguous(), RADIUS)     # HIP: 4 levels x 81 bilinear samples per position
            net, mask, dflow = self.update_block(net, inp, corr, coords1 - coords0)
            coords1 = coords1 + dflow
            up = self._upsample(coords1 - coords0, mask)
        Hp, Wp = up.shape[-2:]
        return up[..., pads[2]:Hp - pads[3], pads[0]:Wp - pads[1]]


_WHEEL = None


def _wheel(device):
    """Middlebury colour wheel, 55 hues (RY 15, YG 6, GC 4, CB 11, BM 13, MR 6), cyclic (56 rows)."""
    global _WHEEL
    if _WHEEL is None:
        hues = [(255, 0, 0), (255, 255, 0), (0, 255, 0), (0, 255, 255), (0, 0, 255), (255, 0, 255), (255, 0, 0)]
        rows = []
        for (a, b), n in zip(zip(hues[:-1], hues[1:]), (15, 6, 4, 11, 13, 6)):
            ta, tb = torch.tensor(a, dtype=torch.float64), torch.tensor(b, dtype=torch.float64)
            rows.append(ta + (tb - ta) * (torch.arange(n, dtype=torch.float64) / n).unsqueeze(1))
        w = torch.cat(rows).to(torch.uint8).to(torch.float32)                # stored as uint8 in the published code
        _WHEEL = torch.cat([w, w[:1]])
    return _WHEEL.to(device)


def flow_to_rgb(flow, flow_max_radius=None):
    """flow f32 [..., 2, H, W] -> RGB f32 [..., 3, H, W] in [0, 1] (flow_utils.flow_to_rgb, bright background): hue =
    flow angle on the cyclic wheel, saturation = radius / the largest radius of the input (or flow_max_radius)."""
    u, v = flow[..., 0, :, :].float(), flow[..., 1, :, :].float()
    radius, angle = torch.hypot(u, v), torch.atan2(v, u)
    mx = radius.max() if flow_max_radius is None else torch.as_tensor(float(flow_max_radius), device=flow.device)
    radius = torch.where(mx > 0, radius / mx.clamp_min(1e-30), radius)
    wheel = _wheel(flow.device)
    angle = torch.where(angle < 0, angle + 2 * math.pi, angle) * ((wheel.shape[0] - 2) / (2 * math.pi))
    lo = torch.floor(angle)
    frac = (angle - lo).unsqueeze(-1)
    hue = wheel[lo.long()] * (1 - frac) + wheel[torch.ceil(angle).long()] * frac
    r = radius.unsqueeze(-1)
    col = torch.where(r > 1, hue / r.clamp_min(1e-30), 255.0 - r * (255.0 - hue))
    return (col.clamp(0, 255) / 255.0).movedim(-1, -3)


def flow_image_uint8(rgb):
    """HWC uint8 image as the reference writes it: np.uint8(flow_rgb * 255) (extract_representations.py:245-249)."""
    return (rgb.permute(1, 2, 0) * 255).to(torch.uint8).cpu().numpy()
