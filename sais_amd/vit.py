"""DINO ViT-S/16 spatial encoder on MI355X — drop-in for the reference's
SAIS/scripts/dino-main/vision_transformer.py (`vit_small`, `VisionTransformer.forward` :209-214,
`get_last_selfattention` :216-223) with the same parameter names / shapes (150-tensor state_dict,
SURVEY App. A), so `dino_deitsmall16_pretrain.pth` loads strictly.

Every block runs as hand-written gfx950 kernels through the C ABI (include/sais_hip.h):
  LN -> [MFMA GEMM qkv] -> [LDS-resident attention] -> [MFMA GEMM proj + residual] ->
  LN -> [MFMA GEMM fc1 + GELU] -> [MFMA GEMM fc2 + residual]
with bf16 MFMA operands, fp32 accumulation, fp32 residual stream and fp32 LN/softmax statistics.
Backward is hand-written too (dX GEMMs on pre-transposed bf16 weight shadows, dW GEMMs with
transposed LDS reads, recompute-from-LSE attention backward); parameter gradients are accumulated
by the kernels directly into the flat gradient buffer that p.grad views.

Deliberate differences from the reference (documented in DESIGN.md): only the ViT-S/16 geometry is supported, at
224 x 224 (197 tokens: every SAIS path and DINO's global crops) and 96 x 96 (37 tokens: DINO's local crops,
main_dino.py:658-663, with interpolate_pos_encoding :174-194 as a fixed bicubic map applied by a HIP kernel).  DropPath (stochastic depth, vision_transformer.py:27-46: train() only, per-sample keep with
rates linspace(0, drop_path_rate, depth); SAIS itself only ever runs the ViT in eval(), extract_representations.py:
279,340,362) is applied in the epilogues of the residual GEMMs as a per-row scale; the keep draws come from Philox
(`drop_path_seed`, this library's stream, not torch's): `last_droppath_scales` holds what a forward used.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .flat import FlatParams

D, NTOK, HEADS, HID, PATCH_K = 384, 197, 6, 1536, 768
_PRUNE_Q = os.environ.get("SAIS_VIT_PRUNE_Q", "1") != "0"     # the CLS-only last block computes q for the CLS rows only
SIDES = {224: 197, 96: 37}          # supported frame sizes -> tokens (the attention kernels are instantiated per count)


def _cubic_taps(t, A=-0.75):
    """The four Keys cubic-convolution weights (A = -0.75, what F.interpolate(mode='bicubic') uses) of the taps at
    floor(x) - 1 .. floor(x) + 2 for fractional offsets t = x - floor(x); t: [n] -> [n, 4]."""
    def near(x):           # |x| <= 1
        return ((A + 2) * x - (A + 3)) * x * x + 1
    def far(x):            # 1 < |x| < 2
        return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A
    return np.stack([far(t + 1), near(t), near(1 - t), far(2 - t)], axis=1)


def pos_interp_matrix(side_in, w, h, patch=16):
    """interpolate_pos_encoding (vision_transformer.py:174-194) as a matrix: [w0 * h0, side_in^2] with
    patch_pos(w, h) = M @ patch_pos(224).  Per axis: F.interpolate(scale_factor=(n0 + 0.1) / side_in, mode='bicubic',
    align_corners=False) samples at (dst + 0.5) / scale - 0.5 with border-clamped taps; the 2-D map is their Kronecker
    product (first axis = the first index of the 14 x 14 grid)."""
    def axis(n0):
        scale = (n0 + 0.1) / side_in
        n_out = int(math.floor(side_in * scale))
        if n_out != n0:
            raise ValueError(f"interpolate_pos_encoding: {n_out} != {n0} (vision_transformer.py:191)")
        x = (np.arange(n_out) + 0.5) / scale - 0.5
        x0 = np.floor(x)
        taps = _cubic_taps(x - x0)
        M = np.zeros((n_out, side_in))
        for k in range(4):
            idx = np.clip(x0.astype(np.int64) + k - 1, 0, side_in - 1)
            np.add.at(M, (np.arange(n_out), idx), taps[:, k])
        return M
    return np.kron(axis(w // patch), axis(h // patch))


class _Attention(nn.Module):
    def __init__(self):
        super().__init__()
        self.qkv = nn.Linear(D, 3 * D, bias=True)
        self.proj = nn.Linear(D, D)


class _Mlp(nn.Module):
    def __init__(self):
        super().__init__()
        self.fc1 = nn.Linear(D, HID)
        self.fc2 = nn.Linear(HID, D)


class _Block(nn.Module):
    def __init__(self):
        super().__init__()
        self.norm1 = nn.LayerNorm(D, eps=1e-6)
        self.attn = _Attention()
        self.norm2 = nn.LayerNorm(D, eps=1e-6)
        self.mlp = _Mlp()


class _PatchEmbed(nn.Module):
    def __init__(self):
        super().__init__()
        self.proj = nn.Conv2d(3, D, kernel_size=16, stride=16)
        self.num_patches = 196
        self.patch_size = 16


def _trunc_normal_(t, std=0.02):
    # utils.trunc_normal_ (dino-main/utils.py:513-551) with a=-2, b=2
    return nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0, b=2.0)


class _ViTFn(torch.autograd.Function):
    """Whole-ViT forward/backward as one autograd node.  Parameter gradients are written by the HIP
    kernels into model.flat.grad (p.grad views) as a side effect; `anchor` only keeps the node alive."""

    @staticmethod
    def forward(ctx, model, x, anchor):
        reps, saved = model._forward_kernels(x, save=True)
        ctx.model, ctx.saved = model, saved
        ctx.x_requires_grad = x.requires_grad
        return reps

    @staticmethod
    def backward(ctx, dreps):
        ctx.model._backward_kernels(ctx.saved, dreps.contiguous())
        ctx.saved = None
        return None, None, None


class VisionTransformer(nn.Module):
    def __init__(self, img_size=[224], patch_size=16, in_chans=3, num_classes=0, embed_dim=384, depth=12,
                 num_heads=6, mlp_ratio=4., qkv_bias=True, drop_path_rate=0., **kwargs):
        super().__init__()
        if (patch_size, in_chans, embed_dim, num_heads, int(embed_dim * mlp_ratio), img_size[0], num_classes) != \
                (16, 3, D, HEADS, HID, 224, 0) or not qkv_bias:
            raise NotImplementedError("the MI355X kernels implement the ViT-S/16 @224 geometry only")
        self.num_features = self.embed_dim = embed_dim
        self.depth = depth
        self.drop_path_rate = drop_path_rate        # identity in eval(); per-sample stochastic depth in train()
        self.drop_path_seed = 0
        # forward() returns norm(x)[:, 0] (:212-214): of the LAST block only the CLS rows are ever read, so its row-local half
        # (proj, norm2, MLP, residual adds) runs on the CLS rows and its attention for the CLS query only; outputs and all
        # parameter gradients are unchanged.  SAIS_VIT_PRUNE_LAST=0 / prune_last_block = False computes every row.
        import os as _os
        self.prune_last_block = _os.environ.get("SAIS_VIT_PRUNE_LAST", "1") != "0"
        # one C call per Block (sais_vit_block_fwd / _bwd) instead of the per-launch Python sequence; SAIS_VIT_BLOCK_CALLS=0 off
        self.block_calls = _os.environ.get("SAIS_VIT_BLOCK_CALLS", "1") != "0"
        self._bp, self._wsbuf = None, {}
        self._rng = None
        self.last_droppath_scales = None
        self.patch_embed = _PatchEmbed()
        self.cls_token = nn.Parameter(torch.zeros(1, 1, D))
        self.pos_embed = nn.Parameter(torch.zeros(1, NTOK, D))
        self.blocks = nn.ModuleList([_Block() for _ in range(depth)])
        self.norm = nn.LayerNorm(D, eps=1e-6)
        self.head = nn.Identity()
        _trunc_normal_(self.pos_embed)
        _trunc_normal_(self.cls_token)
        for m in self.modules():                     # _init_weights, vision_transformer.py:165-172
            if isinstance(m, nn.Linear):
                _trunc_normal_(m.weight)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
        self.flat = None
        self._sig = None
        self._anchor = None
        self._interp = {}                            # frame side -> device f32 [ntok - 1, 196] bicubic map
        self.grad_ready_hook = None                  # callable(lo, hi): flat-grad slice [lo,hi) is final
        self._t_names = []
        for i in range(depth):
            p = f"blocks.{i}."
            self._t_names += [p + "attn.qkv.weight", p + "attn.proj.weight", p + "mlp.fc1.weight", p + "mlp.fc2.weight"]
        self._sentinels = ["cls_token", "patch_embed.proj.weight", "blocks.0.attn.qkv.weight", "norm.weight",
                           f"blocks.{depth - 1}.mlp.fc2.weight"]

    # ------------------------------------------------------------------ engine plumbing
    def _engine(self, device):
        if self.flat is None or not self.flat.intact() or self.flat.device != device:
            self.flat = FlatParams(self, device)
            self._anchor = torch.zeros(1, device=device, requires_grad=True)
            self._sig = None
        sig = self.flat.signature(self._sentinels)
        if sig != self._sig:
            self.flat.refresh_shadows(self._t_names)
            self._sig = self.flat.signature(self._sentinels)
        return self.flat

    def shadows_dirty(self):
        self._sig = None

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._sig = None
        return r

    def sgd_step(self, lr, grad_scale=1.0):
        """Fused vanilla SGD over the whole flat buffer (prepare_model.py:566-567) + shadow refresh."""
        self.flat.sgd_step(lr, grad_scale, self._t_names)
        self._sig = self.flat.signature(self._sentinels)

    def block_grad_range(self, i):
        """[lo, hi) slice of the flat gradient buffer that belongs to block i."""
        f = self.flat
        lo = f.offsets[f"blocks.{i}.norm1.weight"]
        hi = f.offsets[f"blocks.{i + 1}.norm1.weight"] if i + 1 < self.depth else f.offsets["norm.weight"]
        return lo, hi

    # ------------------------------------------------------------------ public API (reference signatures)
    def forward(self, x):
        x = self._check_input(x)
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in (self.cls_token, self.norm.weight))
        self._engine(x.device)
        if need_grad:
            return _ViTFn.apply(self, x, self._anchor)
        return self._forward_kernels(x, save=False)[0]

    def get_last_selfattention(self, x):
        x = self._check_input(x)
        self._engine(x.device)
        return self._forward_kernels(x, save=False, want_last_attn=True)[0]

    def _check_input(self, x):
        if not x.is_cuda:
            raise L.SaisHipError("VisionTransformer.forward needs a device tensor: the HIP path has no CPU fallback")
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != x.shape[3] or x.shape[2] not in SIDES:
            raise ValueError(f"expected [F,3,S,S] with S in {sorted(SIDES)}, got {tuple(x.shape)}")
        return x.contiguous().float()

    def _pos_table(self, side, dev):
        """(positional table f32 [ntok, 384], interpolation map or None) for frames of this side."""
        pos = self.flat.w32("pos_embed").view(NTOK, D)
        if side == 224:
            return pos, None
        Wm = self._interp.get(side)
        if Wm is None or Wm.device != dev:
            Wm = torch.from_numpy(pos_interp_matrix(14, side, side).astype(np.float32)).to(dev)
            self._interp[side] = Wm
        out = torch.empty(SIDES[side], D, dtype=torch.float32, device=dev)
        ops.pos_interp_fwd(Wm, pos, out)
        return out, Wm

    # ------------------------------------------------------------------ forward kernels
    def _forward_kernels(self, img, save, want_last_attn=False):
        """img: one [F,3,S,S] tensor, or a LIST of them with different S (DINO's multi-crop student, utils.py:611-630):
        the groups are then stacked along the token-row axis and every GEMM / LayerNorm of a block runs ONCE over all rows
        (only attention, the embedding glue and the final CLS gather are per group) — 44 160 rows in one launch fill the
        chip where 25 216 + 18 944 in two launches leave workgroup slots empty.  Returns reps f32 [sum F, 384] in group
        order."""
        f = self.flat
        imgs = list(img) if isinstance(img, (list, tuple)) else [img]
        dev = imgs[0].device
        e16 = lambda *s: torch.empty(*s, dtype=torch.bfloat16, device=dev)
        e32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        groups, off, poff, foff = [], 0, 0, 0          # per group: frames, tokens, row / patch-row / frame offsets, interp map
        for im in imgs:
            Fr, ntok = im.shape[0], SIDES[im.shape[-1]]
            groups.append(dict(Fr=Fr, ntok=ntok, off=off, poff=poff, foff=foff, interp=None))
            off, poff, foff = off + Fr * ntok, poff + Fr * (ntok - 1), foff + Fr
        M, NP, Ftot = off, poff, foff
        if want_last_attn and len(groups) > 1:
            raise ValueError("get_last_selfattention takes one resolution")
        patches = e16(NP, PATCH_K)
        x = e32(M, D)
        for im, g in zip(imgs, groups):
            Fr, ntok = g["Fr"], g["ntok"]
            pg = patches[g["poff"]:g["poff"] + Fr * (ntok - 1)]
            xg = x[g["off"]:g["off"] + Fr * ntok].view(Fr, ntok, D)
            ops.patchify(im, pg)
            pos, g["interp"] = self._pos_table(im.shape[-1], dev)
            ops.gemm_nt(pg, f.w("patch_embed.proj.weight").view(D, PATCH_K), L.EPI_PATCH_F32, xg,
                        bias=f.w32("patch_embed.proj.bias"), aux=pos, grp=(ntok - 1, ntok, 1))
            ops.vit_cls_rows(f.w32("cls_token"), pos, xg, Fr, ntok)
        saved = {"patches": patches, "blocks": [], "groups": groups, "M": M, "Ftot": Ftot} if save else None
        # Large M (training step, big extraction batches): the N = 384 GEMMs run on the row-owning kernel with the
        # FOLLOWING LayerNorm in their epilogue (sais_gemm_ln_fwd): proj -> norm2, fc2 -> the next block's norm1.
        # Only block 0's norm1 and the final norm remain stand-alone launches.
        fused = M >= ops.ROW_GEMM_MIN_M
        # DropPath (train mode): row scales keep_f / (1 - p_i) for the 2 x depth residual branches, one Philox draw per
        # frame and branch; branch 2i = attention of block i, 2i + 1 = its MLP.  None in eval() / rate 0.
        dp = None
        if self.training and self.drop_path_rate > 0 and not want_last_attn:
            if self._rng is None or self._rng.device != dev:
                self._rng = ops.rng_state(self.drop_path_seed, dev)
                rates = torch.linspace(0, self.drop_path_rate, self.depth).repeat_interleave(2)      # :150
                self._dp_rates = rates.to(dev, torch.float32)
            parts = []
            for g in groups:                             # one advance + draw per group (what separate passes would do)
                ops.rng_advance(self._rng)
                parts.append(ops.droppath_scales(self._dp_rates, g["Fr"], g["ntok"], self._rng))
            dp = parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)
            self.last_droppath_scales = dp
        if saved is not None:
            saved["dp"] = dp
        xn, qkv, ao, h = e16(M, D), e16(M, 3 * D), e16(M, D), e16(M, HID)
        xn2 = e16(M, D)
        mean1 = rstd1 = None
        # the CLS-only tail runs its [frames, 384] GEMMs on the split-K small-M kernel (SAIS_EPI_RAW_SLABS_F32: M < 8192 only), so a
        # pass with 8192 or more FRAMES computes the last block on every row like the others (pruned=False is
        # recorded in `saved`, the backward follows it)
        prune = self.prune_last_block and not want_last_attn and Ftot < ops.ROW_GEMM_MIN_M
        blockcall = self.block_calls and ops.TIMER is None and len(groups) == 1 and fused
        for i in range(self.depth):
            p = f"blocks.{i}."
            last_attn = want_last_attn and i == self.depth - 1
            if save:
                if prune and i == self.depth - 1:
                    qkv = e16(M, 3 * D)                  # the pruned last block keeps its compact tensors itself
                else:
                    qkv, ao, h = e16(M, 3 * D), e16(M, D), e16(M, HID)
            if i == 0 or not fused:                      # otherwise xn / mean1 / rstd1 came out of the previous fc2
                if save:
                    xn, mean1, rstd1 = e16(M, D), e32(M), e32(M)
                ops.layernorm_fwd(x, M, D, f.w32(p + "norm1.weight"), f.w32(p + "norm1.bias"), 1e-6, y16=xn, mean=mean1,
                                  rstd=rstd1)
            use_block = blockcall and not last_attn and not ops.mlp_fused_enabled(M) and not (prune and i == self.depth - 1)
            if prune and i == self.depth - 1 and len(groups) == 1 and _PRUNE_Q:
                # CLS-only last block: keys and values of every token, the query of the CLS rows only (the other rows' q
                # columns of `qkv` stay unwritten: nothing reads them)
                Wq, bq = f.w(p + "attn.qkv.weight"), f.w32(p + "attn.qkv.bias")
                ops.gemm_nt(xn, Wq[D:], L.EPI_BIAS_BF16, qkv[:, D:], bias=bq[D:])
                g0 = groups[0]
                ops.gemm_nt(xn.view(g0["Fr"], g0["ntok"], D)[:, 0], Wq[:D], L.EPI_BIAS_BF16,
                            qkv.view(g0["Fr"], g0["ntok"], 3 * D)[:, 0, :D], bias=bq[:D])
            elif not use_block:
                ops.gemm_nt(xn, f.w(p + "attn.qkv.weight"), L.EPI_BIAS_BF16, qkv, bias=f.w32(p + "attn.qkv.bias"))
            if prune and i == self.depth - 1:
                reps, tail = self._cls_tail_fwd(f, i, x, xn, mean1, rstd1, qkv, groups, dp, save, e16, e32)
                if save:
                    saved["blocks"].append(tail)
                    saved.update(x_final=tail["x_out"], meanN=tail["meanN"], rstdN=tail["rstdN"], pruned=True)
                return reps, saved
            # One C call per Block (sais_vit_block_fwd: the launches below, sequenced by the library) whenever nothing needs the
            # per-launch view: no HIP-event instrumentation (bench.py's roofline pass), one resolution group, no probabilities
            if use_block:
                g0 = groups[0]
                lse0 = e32(g0["Fr"], HEADS, g0["ntok"]) if save else None
                x_mid = e32(M, D) if save else x
                x_out = e32(M, D) if save else x
                if save:
                    xn2 = e16(M, D)
                mean2, rstd2 = (e32(M), e32(M)) if save else (None, None)
                u = ops.gelu_grad_buffer(M, HID, x.device) if save else None
                nxt = i + 1 < self.depth
                blk = dict(x_in=x, mean1=mean1, rstd1=rstd1, xn1=xn, qkv=qkv, ao=ao, lse=[lse0], x_mid=x_mid, mean2=mean2,
                           rstd2=rstd2, xn2=xn2, dgelu=u, h=h) if save else None
                xn_in = xn
                if nxt and save:
                    xn, mean1, rstd1 = e16(M, D), e32(M), e32(M)
                ops.vit_block_fwd(self._block_params(i), g0["Fr"], g0["ntok"], xn_in, x, qkv, ao, lse0, x_mid, xn2, mean2, rstd2,
                                  h if save else None, u, x_out, xn if nxt else None, mean1 if nxt else None,
                                  rstd1 if nxt else None, None if dp is None else dp[2 * i],
                                  None if dp is None else dp[2 * i + 1], None if save else self._ws(L.OP_VIT_BLOCK_FWD, g0, dev))
                if save:
                    saved["blocks"].append(blk)
                x = x_out
                continue
            lse, probs = [], None
            for g in groups:
                Fr, ntok, lo = g["Fr"], g["ntok"], g["off"]
                lg = e32(Fr, HEADS, ntok) if save else None
                probs = e32(Fr, HEADS, ntok, ntok) if last_attn else None
                ops.vit_attn_fwd(qkv[lo:lo + Fr * ntok], Fr, ao[lo:lo + Fr * ntok], lg, probs, ntok=ntok)
                lse.append(lg)
            if last_attn:
                return probs, None
            x_mid = e32(M, D) if save else x
            if save:
                xn2 = e16(M, D)
            mean2 = e32(M) if save else None
            rstd2 = e32(M) if save else None
            rs_attn = None if dp is None else dp[2 * i]
            rs_mlp = None if dp is None else dp[2 * i + 1]
            if fused:
                ops.gemm_ln_fwd(ao, f.w(p + "attn.proj.weight"), f.w32(p + "attn.proj.bias"), x, x_mid, xn2,
                                f.w32(p + "norm2.weight"), f.w32(p + "norm2.bias"), 1e-6, mean2, rstd2, rowscale=rs_attn)
            else:
                ops.gemm_nt(ao, f.w(p + "attn.proj.weight"), L.EPI_BIAS_RESID_F32, x_mid,
                            bias=f.w32(p + "attn.proj.bias"), aux=x, rowscale=rs_attn)
                ops.layernorm_fwd(x_mid, M, D, f.w32(p + "norm2.weight"), f.w32(p + "norm2.bias"), 1e-6, y16=xn2,
                                  mean=mean2, rstd=rstd2)
            u = ops.gelu_grad_buffer(M, HID, x.device) if save else None
            x_out = e32(M, D) if save else x
            blk = dict(x_in=x, mean1=mean1, rstd1=rstd1, xn1=xn, qkv=qkv, ao=ao, lse=lse, x_mid=x_mid, mean2=mean2,
                       rstd2=rstd2, xn2=xn2, dgelu=u, h=h) if save else None
            nxt = fused and i + 1 < self.depth
            if nxt and save:
                xn, mean1, rstd1 = e16(M, D), e32(M), e32(M)
            q = f"blocks.{i + 1}." if nxt else None
            if ops.mlp_fused_enabled(M):
                # the whole MLP branch in ONE launch: fc1 + GELU (+ GELU') -> fc2 + residual (+ the next block's norm1); h is
                # written for the backward pass but never read back, and in inference it is not materialised at all
                ops.mlp_fwd(xn2, f.w(p + "mlp.fc1.weight"), f.w32(p + "mlp.fc1.bias"), f.w(p + "mlp.fc2.weight"),
                            f.w32(p + "mlp.fc2.bias"), x_mid, x_out, h=h if save else None, g=u,
                            xn_out=xn if nxt else None, gamma=f.w32(q + "norm1.weight") if nxt else None,
                            beta=f.w32(q + "norm1.bias") if nxt else None, eps=1e-6, mean=mean1 if nxt else None,
                            rstd=rstd1 if nxt else None, rowscale=rs_mlp)
            else:
                ops.gemm_nt(xn2, f.w(p + "mlp.fc1.weight"), ops.epi_gelu_grad(M) if save else L.EPI_BIAS_GELU_BF16, h,
                            bias=f.w32(p + "mlp.fc1.bias"), out2=u)
                if nxt:
                    ops.gemm_ln_fwd(h, f.w(p + "mlp.fc2.weight"), f.w32(p + "mlp.fc2.bias"), x_mid, x_out, xn,
                                    f.w32(q + "norm1.weight"), f.w32(q + "norm1.bias"), 1e-6, mean1, rstd1, rowscale=rs_mlp)
                else:
                    ops.gemm_nt(h, f.w(p + "mlp.fc2.weight"), L.EPI_BIAS_RESID_F32, x_out, bias=f.w32(p + "mlp.fc2.bias"),
                                aux=x_mid, rowscale=rs_mlp)
            if save:
                saved["blocks"].append(blk)
            x = x_out
        reps = e32(Ftot, D)
        meanN = e32(Ftot) if save else None
        rstdN = e32(Ftot) if save else None
        for g in groups:                                 # final norm on the CLS rows only (row stride = ntok * 384)
            Fr, ntok, lo, fo = g["Fr"], g["ntok"], g["off"], g["foff"]
            ops.layernorm_fwd(x[lo:lo + Fr * ntok], Fr, ntok * D, f.w32("norm.weight"), f.w32("norm.bias"), 1e-6,
                              y32=reps[fo:fo + Fr], mean=None if meanN is None else meanN[fo:fo + Fr],
                              rstd=None if rstdN is None else rstdN[fo:fo + Fr])
        if save:
            saved.update(x_final=x, meanN=meanN, rstdN=rstdN)
        return reps, saved

    def _block_params(self, i):
        """ctypes parameter blocks of the Blocks (pointers into the flat buffers), rebuilt when those are."""
        f = self.flat
        key = (f.flat.data_ptr(), f.grad.data_ptr(), f.w16.data_ptr(), 0 if f.wt_buf is None else f.wt_buf.data_ptr())
        if self._bp is None or self._bp[0] != key:
            self._bp = (key, [ops.vit_block_params(f, j, self.depth) for j in range(self.depth)])
        return self._bp[1][i]

    def _ws(self, op, grp, dev, slot=0):
        """Scratch for a block-level call (sais_workspace_bytes), kept per (op, shape, slot) outside hipGraph pools.  slot: the
        position of a backward call among those whose weight gradients are still to be launched (their scratch must survive)."""
        if torch.cuda.is_current_stream_capturing():
            return ops.block_workspace(op, grp["Fr"], grp["ntok"], dev)
        key = (op, grp["Fr"], grp["ntok"], str(dev), slot)
        if key not in self._wsbuf:
            self._wsbuf[key] = ops.block_workspace(op, grp["Fr"], grp["ntok"], dev)
        return self._wsbuf[key]

    def _cls_tail_fwd(self, f, i, x, xn1, mean1, rstd1, qkv, groups, dp, save, e16, e32):
        """The last block from its qkv on, restricted to what forward() returns (the CLS rows): attention for the CLS query
        (sais_vit_attn_cls_fwd, one launch per resolution group), then proj + residual, norm2, fc1 + GELU, fc2 + residual and
        the final norm on [frames, 384] tensors (all groups together).  The residual input is the strided view x[::ntok]."""
        p = f"blocks.{i}."
        Ftot = sum(g["Fr"] for g in groups)
        ao = e16(Ftot, D)
        for g in groups:
            Fr, ntok, lo, fo = g["Fr"], g["ntok"], g["off"], g["foff"]
            ops.vit_attn_cls_fwd(qkv[lo:lo + Fr * ntok], Fr, ao[fo:fo + Fr], ntok)
        cls_rows = lambda t, w: [t[g["off"]:g["off"] + g["Fr"] * g["ntok"]].view(g["Fr"], g["ntok"], *w)[:, 0] for g in groups]
        one = len(groups) == 1
        x_cls = cls_rows(x, (D,))[0] if one else torch.cat(cls_rows(x, (D,)))
        rs_attn = None if dp is None else torch.cat(cls_rows(dp[2 * i], ())).contiguous()
        rs_mlp = None if dp is None else torch.cat(cls_rows(dp[2 * i + 1], ())).contiguous()
        x_mid, xn2, x_out = e32(Ftot, D), e16(Ftot, D), e32(Ftot, D)
        mean2, rstd2 = (e32(Ftot), e32(Ftot)) if save else (None, None)
        # [frames, 384] outputs: 6 tiles of 128 x 128 — the K loop is cut over workgroups (deterministic split-K)
        ops.gemm_nt_splitk(ao, f.w(p + "attn.proj.weight"), 6, bias=f.w32(p + "attn.proj.bias"), rowscale=rs_attn, aux=x_cls,
                           out32=x_mid)
        ops.layernorm_fwd(x_mid, Ftot, D, f.w32(p + "norm2.weight"), f.w32(p + "norm2.bias"), 1e-6, y16=xn2, mean=mean2,
                          rstd=rstd2)
        h, u = e16(Ftot, HID), (ops.gelu_grad_buffer(Ftot, HID, x.device) if save else None)
        ops.gemm_nt(xn2, f.w(p + "mlp.fc1.weight"), ops.epi_gelu_grad(Ftot) if save else L.EPI_BIAS_GELU_BF16, h,
                    bias=f.w32(p + "mlp.fc1.bias"), out2=u)
        ops.gemm_nt_splitk(h, f.w(p + "mlp.fc2.weight"), 12, bias=f.w32(p + "mlp.fc2.bias"), rowscale=rs_mlp, aux=x_mid,
                           out32=x_out)
        reps = e32(Ftot, D)
        meanN, rstdN = (e32(Ftot), e32(Ftot)) if save else (None, None)
        ops.layernorm_fwd(x_out, Ftot, D, f.w32("norm.weight"), f.w32("norm.bias"), 1e-6, y32=reps, mean=meanN, rstd=rstdN)
        tail = dict(cls=True, x_in=x, mean1=mean1, rstd1=rstd1, xn1=xn1, qkv=qkv, ao=ao, x_mid=x_mid, mean2=mean2, rstd2=rstd2,
                    xn2=xn2, dgelu=u, h=h, rs_attn=rs_attn, rs_mlp=rs_mlp, x_out=x_out, meanN=meanN, rstdN=rstdN) if save else None
        return reps, tail

    def _cls_tail_bwd(self, f, saved, dreps, dx, dxa, dqkv, fused, defer=None):
        """Backward of _cls_tail_fwd: the gradient enters on the CLS rows only.  Everything row-local stays on [frames, 384]
        tensors; the attention backward (sais_vit_attn_cls_bwd) writes the whole dqkv (dk, dv of every token, dq of the CLS
        rows, zeros elsewhere); the dX of qkv + norm1's backward then writes EVERY row of dx / dxa, taking the residual-stream
        gradient from the compact CLS tensor (dres_period) — no zero-filled [M, 384] buffer, no full-size cast.  With several
        resolution groups the two token-count-dependent launches run once per group on its rows."""
        i = self.depth - 1
        p = f"blocks.{i}."
        s = saved["blocks"][i]
        groups, M = saved["groups"], saved["M"]
        Ftot = sum(g["Fr"] for g in groups)
        dev = dreps.device
        e16 = lambda *sh: torch.empty(*sh, dtype=torch.bfloat16, device=dev)
        dx_c = torch.empty(Ftot, D, dtype=torch.float32, device=dev)
        ops.layernorm_bwd(s["x_out"], D, s["meanN"], s["rstdN"], f.w32("norm.weight"), Ftot, dy32=dreps, dx32=dx_c,
                          dgamma=f.g("norm.weight"), dbeta=f.g("norm.bias"))
        if self.grad_ready_hook:
            self.grad_ready_hook(f.offsets["norm.weight"], f.numel)
        dxa_c, du, dxn, dxb_c, dao = e16(Ftot, D), e16(Ftot, HID), e16(Ftot, D), e16(Ftot, D), e16(Ftot, D)
        if s["rs_mlp"] is None:
            ops.cast_bf16(dx_c, dxa_c)
        else:
            ops.cast_bf16_rows(dx_c, s["rs_mlp"], dxa_c)
        ops.gemm_nt(dxa_c, f.wt16[p + "mlp.fc2.weight"], ops.epi_mul(Ftot), du, aux=s["dgelu"])
        ops.gemm_nt_splitk(du, f.wt16[p + "mlp.fc1.weight"], 12, out16=dxn)
        ops.layernorm_bwd(s["x_mid"], D, s["mean2"], s["rstd2"], f.w32(p + "norm2.weight"), Ftot, dy16=dxn, dres=dx_c, dx32=dx_c,
                          dx16=dxb_c, dgamma=f.g(p + "norm2.weight"), dbeta=f.g(p + "norm2.bias"), rowscale16=s["rs_attn"])
        ops.gemm_nt_splitk(dxb_c, f.wt16[p + "attn.proj.weight"], 6, out16=dao)
        for g in groups:
            Fr, ntok, lo, fo = g["Fr"], g["ntok"], g["off"], g["foff"]
            ops.vit_attn_cls_bwd(s["qkv"][lo:lo + Fr * ntok], dao[fo:fo + Fr], Fr, dqkv[lo:lo + Fr * ntok], ntok)
        compact = [
            (dxa_c, s["h"], f.g(p + "mlp.fc2.weight"), f.g(p + "mlp.fc2.bias")),
            (du, s["xn2"], f.g(p + "mlp.fc1.weight"), f.g(p + "mlp.fc1.bias")),
            (dxb_c, s["ao"], f.g(p + "attn.proj.weight"), f.g(p + "attn.proj.bias"))]
        gW, gb = f.g(p + "attn.qkv.weight"), f.g(p + "attn.qkv.bias")
        if len(groups) == 1 and _PRUNE_Q:      # dq is zero off the CLS rows: its weight gradient is a [frames, 384] GEMM too
            Fr, ntok = groups[0]["Fr"], groups[0]["ntok"]
            compact.append((dqkv.view(Fr, ntok, 3 * D)[:, 0, :D], s["xn1"].view(Fr, ntok, D)[:, 0], gW[:D], gb[:D]))
            ops.gemm_tn_grouped(compact, Ftot)
            kv = (dqkv[:, D:], s["xn1"], gW[D:], gb[D:])
            if defer is not None:                     # rides in the first grouped launch of the blocks below (4 more tiles)
                defer.append(kv)
            else:
                ops.gemm_tn_grouped([kv], M)
        else:
            ops.gemm_tn_grouped(compact, Ftot)
            ops.gemm_tn_grouped([(dqkv, s["xn1"], gW, gb)], M)
        dp = saved.get("dp")
        rs_prev = None if dp is None or i == 0 else dp[2 * (i - 1) + 1]
        if fused and all(g["Fr"] * g["ntok"] >= ops.ROW_GEMM_MIN_M for g in groups):
            for g in groups:
                Fr, ntok, lo, fo = g["Fr"], g["ntok"], g["off"], g["foff"]
                hi = lo + Fr * ntok
                ops.gemm_ln_bwd(dqkv[lo:hi], f.wt16[p + "attn.qkv.weight"], s["x_in"][lo:hi], s["mean1"][lo:hi], s["rstd1"][lo:hi],
                                f.w32(p + "norm1.weight"), dres=dx_c[fo:fo + Fr], dres_period=ntok, dx32=dx[lo:hi],
                                dx16=dxa[lo:hi], dgamma=f.g(p + "norm1.weight"), dbeta=f.g(p + "norm1.bias"),
                                rowscale16=None if rs_prev is None else rs_prev[lo:hi],
                                xn16=s["xn1"][lo:hi], beta=f.w32(p + "norm1.bias"))
        else:                                         # small M: scatter the CLS gradient into a zeroed residual-stream gradient
            dx.zero_()
            for g in groups:
                Fr, ntok, lo, fo = g["Fr"], g["ntok"], g["off"], g["foff"]
                dx[lo:lo + Fr * ntok].view(Fr, ntok, D)[:, 0].copy_(dx_c[fo:fo + Fr])
            dxn_full = e16(M, D)
            ops.gemm_nt(dqkv, f.wt16[p + "attn.qkv.weight"], L.EPI_BIAS_BF16, dxn_full)
            ops.layernorm_bwd(s["x_in"], D, s["mean1"], s["rstd1"], f.w32(p + "norm1.weight"), M, dy16=dxn_full, dres=dx,
                              dx32=dx, dx16=dxa, dgamma=f.g(p + "norm1.weight"), dbeta=f.g(p + "norm1.bias"),
                              rowscale16=rs_prev)
        saved["blocks"][i] = None
        if self.grad_ready_hook:                      # (with a hook nothing of this block is deferred: _backward_kernels)
            self.grad_ready_hook(*self.block_grad_range(i))

    # ------------------------------------------------------------------ backward kernels
    def _backward_kernels(self, saved, dreps):
        f = self.flat
        f.attach_grads()
        dev = dreps.device
        groups, M = saved["groups"], saved["M"]
        e16 = lambda *s: torch.empty(*s, dtype=torch.bfloat16, device=dev)
        pruned = bool(saved.get("pruned"))
        fused = M >= ops.ROW_GEMM_MIN_M
        dxa, dxb = e16(M, D), e16(M, D)            # bf16 copies of the residual-stream gradient (block input / mid)
        # Weight gradients of G blocks in ONE grouped launch (LABNOTES R6.8): a block's dW needs ten M-splits to fill the chip with
        # its 24 tiles, each split writing a partial tile that a second launch sums; 24 G tiles need 10 / G.  The operands of a
        # deferred block (its du / d(mid) / dqkv, the bf16 gradient that entered it, the saved activations) stay alive until then.
        G = 1
        # a data-parallel hook that all-reduces the blocks' gradients while the backward goes on (an idle one — GradSync in a world
        # of one — carries active = False)
        hooked = self.grad_ready_hook is not None and getattr(self.grad_ready_hook, "active", True)
        if fused and M % 32 == 0 and not ops.mlp_fused_enabled(M):      # (several resolution groups: the per-launch path below)
            G = ops.dw_group(dp_hooks=hooked)
        pend_extra = []
        if pruned:                                  # every row of dx is WRITTEN by the last block's dX qkv + norm1' kernel
            dx = torch.empty(M, D, dtype=torch.float32, device=dev)
            dqkv0 = e16(M, 3 * D)
            self._cls_tail_bwd(f, saved, dreps, dx, dxa, dqkv0, fused,
                               defer=pend_extra if G > 1 and not hooked else None)
        else:
            dx = torch.zeros(M, D, dtype=torch.float32, device=dev)
        for g in ([] if pruned else groups):
            Fr, ntok, lo, fo = g["Fr"], g["ntok"], g["off"], g["foff"]
            ops.layernorm_bwd(saved["x_final"][lo:lo + Fr * ntok], ntok * D, saved["meanN"][fo:fo + Fr],
                              saved["rstdN"][fo:fo + Fr], f.w32("norm.weight"), Fr, dy32=dreps[fo:fo + Fr],
                              dx32=dx[lo:lo + Fr * ntok], lddx32=ntok * D, dgamma=f.g("norm.weight"),
                              dbeta=f.g("norm.bias"))
        dp = saved.get("dp")
        # with DropPath the gradient that enters a branch is s dx (the residual stream keeps dx): the bf16 copies carry the
        # NEXT branch's scale — dxa feeds an MLP branch (2i + 1), dxb an attention branch (2i)
        if pruned:
            pass
        elif dp is None:
            ops.cast_bf16(dx, dxa)
        else:
            ops.cast_bf16_rows(dx, dp[2 * (self.depth - 1) + 1], dxa)
        dxn, dao, du = (None if fused else e16(M, D)), e16(M, D), e16(M, HID)
        dqkv = dqkv0 if pruned else e16(M, 3 * D)
        if self.grad_ready_hook and not pruned:
            self.grad_ready_hook(f.offsets["norm.weight"], f.numel)
        pend_calls, pend_items, pend_blocks = [], [], []

        def flush_dw():
            if pend_calls:
                ops.vit_blocks_dw([c[:3] for c in pend_calls], extra=pend_extra)
                pend_extra.clear()
            if pend_items:
                ops.gemm_tn_grouped(pend_items + pend_extra, M)
                pend_extra.clear()
            for j in pend_blocks:
                saved["blocks"][j] = None
                if self.grad_ready_hook:              # the block's gradient slice is final: DP all-reduce may start
                    self.grad_ready_hook(*self.block_grad_range(j))
            pend_calls.clear(); pend_items.clear(); pend_blocks.clear()

        for i in reversed(range(self.depth - 1 if pruned else self.depth)):
            p = f"blocks.{i}."
            s = saved["blocks"][i]
            rs_attn = None if dp is None else dp[2 * i]                        # this block's attention branch
            rs_prev = None if dp is None or i == 0 else dp[2 * (i - 1) + 1]     # the MLP branch of block i - 1
            if self.block_calls and ops.TIMER is None and len(groups) == 1 and fused and not ops.mlp_fused_enabled(M):
                g0 = groups[0]
                if G > 1:
                    dxa_out = e16(M, D)
                    ws = self._ws(L.OP_VIT_BLOCK_BWD, g0, dev, slot=len(pend_calls))
                    a = ops.vit_block_bwd(self._block_params(i), g0["Fr"], g0["ntok"], s, dx, dxa, dxa_out, rs_attn, rs_prev,
                                          s["lse"][0], ws, defer_dw=True)
                    pend_calls.append((self._block_params(i), a, ws, dxa, s))          # dxa, s: kept alive for the launch
                    pend_blocks.append(i)
                    dxa = dxa_out
                    if len(pend_blocks) == G or i == 0:
                        flush_dw()
                    continue
                ops.vit_block_bwd(self._block_params(i), g0["Fr"], g0["ntok"], s, dx, dxa, dxa, rs_attn, rs_prev, s["lse"][0],
                                  self._ws(L.OP_VIT_BLOCK_BWD, g0, dev))
                saved["blocks"][i] = None
                if self.grad_ready_hook:
                    self.grad_ready_hook(*self.block_grad_range(i))
                continue
            if G > 1:                                 # per-launch path with deferred dW: this block's gradient tensors are its own
                du, dqkv, dxb = e16(M, HID), e16(M, 3 * D), e16(M, D)
            # MLP branch
            if fused and ops.mlp_fused_enabled(M):    # dX of fc2 x GELU' -> dX of fc1 -> norm2's backward: ONE launch
                ops.mlp_bwd(dxa, f.wt16[p + "mlp.fc2.weight"], s["dgelu"], f.wt16[p + "mlp.fc1.weight"], du, s["x_mid"],
                            s["mean2"], s["rstd2"], f.w32(p + "norm2.weight"), dres=dx, dx32=dx, dx16=dxb,
                            dgamma=f.g(p + "norm2.weight"), dbeta=f.g(p + "norm2.bias"), rowscale16=rs_attn)
            elif fused:       # dX of fc1 with norm2's backward (+ residual gradient) in its epilogue
                ops.gemm_nt(dxa, f.wt16[p + "mlp.fc2.weight"], ops.epi_mul(M), du, aux=s["dgelu"])
                ops.gemm_ln_bwd(du, f.wt16[p + "mlp.fc1.weight"], s["x_mid"], s["mean2"], s["rstd2"],
                                f.w32(p + "norm2.weight"), dres=dx, dx32=dx, dx16=dxb, dgamma=f.g(p + "norm2.weight"),
                                dbeta=f.g(p + "norm2.bias"), rowscale16=rs_attn, xn16=s["xn2"], beta=f.w32(p + "norm2.bias"))
            else:
                ops.gemm_nt(dxa, f.wt16[p + "mlp.fc2.weight"], ops.epi_mul(M), du, aux=s["dgelu"])
                ops.gemm_nt(du, f.wt16[p + "mlp.fc1.weight"], L.EPI_BIAS_BF16, dxn)
                ops.layernorm_bwd(s["x_mid"], D, s["mean2"], s["rstd2"], f.w32(p + "norm2.weight"), M, dy16=dxn, dres=dx,
                                  dx32=dx, dx16=dxb, dgamma=f.g(p + "norm2.weight"), dbeta=f.g(p + "norm2.bias"),
                                  rowscale16=rs_attn)
            # attention branch
            ops.gemm_nt(dxb, f.wt16[p + "attn.proj.weight"], L.EPI_BIAS_BF16, dao)
            for g, lg in zip(groups, s["lse"]):
                Fr, ntok, lo = g["Fr"], g["ntok"], g["off"]
                hi = lo + Fr * ntok
                ops.vit_attn_bwd(s["qkv"][lo:hi], dao[lo:hi], s["ao"][lo:hi], lg, None, Fr, dqkv[lo:hi], ntok=ntok)
            # all four weight / bias gradients of the block in one launch.  (A side stream for this launch was measured
            # in round 1: 20.4 vs 19.6 ms/step — both kernels fill the chip — and removed.)
            dw = [(dxa, s["h"], f.g(p + "mlp.fc2.weight"), f.g(p + "mlp.fc2.bias")),
                  (du, s["xn2"], f.g(p + "mlp.fc1.weight"), f.g(p + "mlp.fc1.bias")),
                  (dxb, s["ao"], f.g(p + "attn.proj.weight"), f.g(p + "attn.proj.bias")),
                  (dqkv, s["xn1"], f.g(p + "attn.qkv.weight"), f.g(p + "attn.qkv.bias"))]
            if G > 1:
                pend_items += dw
                pend_blocks.append(i)
                dxa = e16(M, D)                       # norm1's backward writes the NEXT block's bf16 gradient beside this one's
            else:
                ops.gemm_tn_grouped(dw, M)
            if fused:         # dX of qkv with norm1's backward in its epilogue
                ops.gemm_ln_bwd(dqkv, f.wt16[p + "attn.qkv.weight"], s["x_in"], s["mean1"], s["rstd1"],
                                f.w32(p + "norm1.weight"), dres=dx, dx32=dx, dx16=dxa, dgamma=f.g(p + "norm1.weight"),
                                dbeta=f.g(p + "norm1.bias"), rowscale16=rs_prev, xn16=s["xn1"], beta=f.w32(p + "norm1.bias"))
            else:
                ops.gemm_nt(dqkv, f.wt16[p + "attn.qkv.weight"], L.EPI_BIAS_BF16, dxn)
                ops.layernorm_bwd(s["x_in"], D, s["mean1"], s["rstd1"], f.w32(p + "norm1.weight"), M, dy16=dxn, dres=dx,
                                  dx32=dx, dx16=dxa, dgamma=f.g(p + "norm1.weight"), dbeta=f.g(p + "norm1.bias"),
                                  rowscale16=rs_prev)
            if G > 1:
                if len(pend_blocks) == G or i == 0:
                    flush_dw()
                continue
            saved["blocks"][i] = None
            if self.grad_ready_hook:                  # the block's gradient slice is final: DP all-reduce may start
                self.grad_ready_hook(*self.block_grad_range(i))
        flush_dw()
        if pend_extra:                                # (no block below the CLS-only one: depth 1)
            ops.gemm_tn_grouped(pend_extra, M)
        dpatch = e16(saved["patches"].shape[0], D)
        for g in groups:
            Fr, ntok, lo, po = g["Fr"], g["ntok"], g["off"], g["poff"]
            dxg, dpg = dx[lo:lo + Fr * ntok], dpatch[po:po + Fr * (ntok - 1)]
            if g["interp"] is None:
                ops.vit_embed_bwd(dxg, Fr, f.g("cls_token"), f.g("pos_embed"), dpg)
            else:                                   # through the transpose of the bicubic map (autograd of :174-194)
                dpos = torch.zeros(ntok, D, dtype=torch.float32, device=dev)
                ops.vit_embed_bwd(dxg, Fr, f.g("cls_token"), dpos, dpg, ntok=ntok)
                ops.pos_interp_bwd(g["interp"], dpos, f.g("pos_embed").view(NTOK, D))
        # (the ping-pong 128 x 384 dW kernel is no better here: 6 wide tiles x 42 row splits, 77 us against 65-68 us)
        ops.gemm_tn(dpatch, saved["patches"], f.g("patch_embed.proj.weight").view(D, PATCH_K),
                    f.g("patch_embed.proj.bias"))
        if self.grad_ready_hook:
            self.grad_ready_hook(0, f.offsets["blocks.0.norm1.weight"])


def vit_small(patch_size=16, **kwargs):
    """vision_transformer.py:243-247."""
    return VisionTransformer(patch_size=patch_size, embed_dim=384, depth=kwargs.pop("depth", 12), num_heads=6,
                             mlp_ratio=4, qkv_bias=True, **kwargs)
