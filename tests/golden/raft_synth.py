"""Seeded synthetic RAFT weights with the published model's state-dict keys and shapes (no checkpoint exists offline)."""
import torch


def raft_state_dict(seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(name, co, ci, kh, kw):
        fan = ci * kh * kw
        sd[name + ".weight"] = torch.randn(co, ci, kh, kw, generator=g) * (scale * (2.0 / fan) ** 0.5)
        sd[name + ".bias"] = torch.randn(co, generator=g) * 0.05

    def bn(name, c):
        sd[name + ".weight"] = 1 + 0.1 * torch.randn(c, generator=g)
        sd[name + ".bias"] = 0.1 * torch.randn(c, generator=g)
        sd[name + ".running_mean"] = 0.1 * torch.randn(c, generator=g)
        sd[name + ".running_var"] = 1 + 0.2 * torch.rand(c, generator=g)
        sd[name + ".num_batches_tracked"] = torch.tensor(1)

    for pre, kind in (("fnet.", "instance"), ("cnet.", "batch")):
        conv(pre + "conv1", 64, 3, 7, 7)
        if kind == "batch":
            bn(pre + "norm1", 64)
        cin = 64
        for st, c in ((1, 64), (2, 96), (3, 128)):
            for b in (0, 1):
                p = f"{pre}layer{st}.{b}."
                conv(p + "conv1", c, cin if b == 0 else c, 3, 3)
                conv(p + "conv2", c, c, 3, 3)
                if kind == "batch":
                    bn(p + "norm1", c), bn(p + "norm2", c)
                if b == 0 and st > 1:
                    conv(p + "downsample.0", c, cin, 1, 1)
                    if kind == "batch":
                        bn(p + "norm3", c)
            cin = c
        conv(pre + "conv2", 256, 128, 1, 1)
    u = "update_block."
    conv(u + "encoder.convc1", 256, 324, 1, 1), conv(u + "encoder.convc2", 192, 256, 3, 3)
    conv(u + "encoder.convf1", 128, 2, 7, 7), conv(u + "encoder.convf2", 64, 128, 3, 3), conv(u + "encoder.conv", 126, 256, 3, 3)
    for k, (kh, kw) in (("1", (1, 5)), ("2", (5, 1))):
        for n in "zrq":
            conv(u + f"gru.conv{n}{k}", 128, 384, kh, kw)
    conv(u + "flow_head.conv1", 256, 128, 3, 3), conv(u + "flow_head.conv2", 2, 256, 3, 3)
    conv(u + "mask.0", 256, 128, 3, 3), conv(u + "mask.2", 576, 256, 1, 1)
    return sd


def frame_pair(seed, H, W, shift=(3, -2)):
    """Two frames in [0, 1]: smooth random texture and the same texture displaced by `shift` pixels (x, y)."""
    g = torch.Generator().manual_seed(seed)
    base = torch.nn.functional.interpolate(torch.rand(1, 3, H // 4 + 8, W // 4 + 8, generator=g), scale_factor=4, mode="bicubic",
                                           align_corners=False).clamp(0, 1)
    a = base[:, :, 16:16 + H, 16:16 + W]
    b = base[:, :, 16 - shift[1]:16 - shift[1] + H, 16 - shift[0]:16 - shift[0] + W]
    return a.contiguous(), b.contiguous()
