"""Temporal TransformerEncoder + prototype head of SAIS on MI355X — drop-in for the hot-path slice of
the reference's `fullModel` (SAIS/scripts/prepare_model.py:18-488): data_type='reps',
encoder_type='ViT', task='Prototypes', self_attention=True (what main.sh:27 runs).

Same constructor / forward signatures, same parameter names and shapes (4118 keys
incl. the 2x2000 single-row position parameters and the never-used clip/MIL tensors), so
`params.zip` loads strictly once the `module.` prefix is stripped and `encoder.*` ballast dropped.

forward(x, f, xlens, flens, task, xpad, fpad, domains) -> (emb [B,256], attn [B,T+1,T+1])
(list inputs = test-time-augmentation versions -> (list of embs, attn of version 0), :331-346).

Kernels: prepare (CLS + positions) -> 4 x [MFMA in_proj -> LDS masked attention (+ head-averaged map
on the last layer) -> MFMA out_proj + residual -> LN -> MFMA linear1 + ReLU -> MFMA linear2 + residual
-> LN] -> fused ReLU/CLS/two-stream-add/ReLU/Linear(384->256) head.  Backward is hand-written as well.
The MFMA GEMMs here are the fp32-operand "bf16x3" variant (sais_gemm_nt_f32).

Deliberate differences: inputs are never mutated (the reference does `x += pos` in place,
:192, and `rgb += flow`, :412); dropout (p=0.1, the nn.TransformerEncoderLayer default, train() only) is applied at its four sites per layer with
Philox masks of this library's own stream (`dropout_p`, `dropout_seed`; `dropout_p = 0` switches it off);
ClassificationHead / R3D / raw branches are out of scope and raise; task 'MIL' runs in the inference direction (its
training raises inside the reference), multi-domain models ('+' in the domain) select linear / linearB per sample.  Inputs [B, nsnippets, T, 384]: every
(clip, snippet) pair is one sequence of the encoder, the head averages the ReLU'd CLS rows over the snippets of a clip
(:381-382) and the returned attention map is [B*nsnippets, T+1, T+1], as in the reference.
"""
import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .flat import FlatParams

import os as _os

_LAYER_CALLS = _os.environ.get('SAIS_TEMPORAL_LAYER_CALLS', '1') != '0'      # one C call per encoder layer and direction
_PREFETCH = _os.environ.get('SAIS_TEMPORAL_PREFETCH', '0') == '1'        # measured: no net gain (LABNOTES R4.3): off
_TTA_MERGE = _os.environ.get('SAIS_TTA_MERGE', '1') != '0'            # inference: all TTA versions and both streams in ONE encoder pass
_DW_DEFER = _os.environ.get('SAIS_TEMPORAL_DW_DEFER', '1') != '0'       # all layers' weight gradients in ONE launch after the dX chain
D, TH, FF, EMB, NPOS = 384, 4, 2048, 256, 2000


def _make_encoder(rep_dim):
    layer = nn.TransformerEncoderLayer(d_model=rep_dim, nhead=TH)       # prepare_model.py:75
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return nn.TransformerEncoder(layer, num_layers=4)               # :76 (deep-copies one layer)


class _TemporalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, f, xpad, fpad, anchor, second=None):
        emb, attn, imp, saved = model._forward_kernels(x, f, xpad, fpad, save=True, second=second)
        ctx.model, ctx.saved = model, saved
        ctx.needs = (x is not None and x.requires_grad, f is not None and f.requires_grad)
        ctx.mark_non_differentiable(attn)
        if imp is None:
            imp = emb.new_zeros(1)
            ctx.mark_non_differentiable(imp)
        return emb, attn, imp

    @staticmethod
    def backward(ctx, demb, _dattn, dimp):
        dimp = dimp.contiguous() if ctx.model.importance_loss and dimp is not None else None
        dx, df = ctx.model._backward_kernels(ctx.saved, demb.contiguous(), ctx.needs, dimp)
        ctx.saved = None
        return None, dx, df, None, None, None, None


class fullModel(nn.Module):
    def __init__(self, data_type='raw', nclasses=2, domain='NH_02', rep_dim=512, encoder_type='R3D',
                 modalities='RGB-Flow', encoder_depth=18, load_pretrained_params=True, freeze_encoder_params=True,
                 self_attention=True, importance_loss=False):
        super().__init__()
        if data_type != 'reps' or encoder_type != 'ViT' or rep_dim != D or not self_attention:
            raise NotImplementedError("MI355X hot path = fullModel('reps', ..., 384, 'ViT', self_attention=True); "
                                      "raw / R3D / I3D branches are out of scope (SURVEY §2)")
        if importance_loss and modalities == 'Flow':
            raise NotImplementedError("importance head with the Flow-only stream: the reference itself fails there "
                                      "(full_snip_sequence is undefined, prepare_model.py:419-421)")
        if modalities not in ('RGB', 'Flow', 'RGB-Flow'):
            raise ValueError(modalities)
        # registration order mirrors the reference so state_dict ordering matches too
        self.linear = nn.Linear(rep_dim, EMB)
        if '+' in domain:
            self.linearB = nn.Linear(rep_dim, EMB)
        self.linear2 = nn.Linear(EMB, 3)                                 # :50 (dead on this path, App. B.5)
        if importance_loss:
            self.importance_function = nn.Linear(rep_dim, 1)             # :55-56 (a13, optional -il)
        self.frame_cls = nn.Parameter(torch.rand(1, rep_dim))
        self.clip_cls = nn.Parameter(torch.rand(1, rep_dim))
        # filled one key at a time (as the reference does, :64-69): ParameterDict(dict) would SORT the keys
        # lexicographically, and rows 0..T-1 must be consecutive in the flat buffer for the prepare kernel
        self.frame_pos_embeddings = nn.ParameterDict()
        self.clip_pos_embeddings = nn.ParameterDict()
        for table in (self.frame_pos_embeddings, self.clip_pos_embeddings):
            for i in range(NPOS):
                table[str(i)] = nn.Parameter(torch.rand(1, rep_dim))
        self.transEncoderFrame = _make_encoder(rep_dim)
        self.transEncoderClip = _make_encoder(rep_dim)
        self.attentionA = nn.Linear(rep_dim, EMB)
        self.attentionB = nn.Linear(rep_dim, EMB)
        self.attentionModules = nn.ModuleDict({str(c): nn.Linear(EMB, 1) for c in range(3)})
        self.finalModules = nn.ModuleDict({str(c): nn.Linear(rep_dim, 1) for c in range(3)})
        self.nclasses, self.domain, self.rep_dim = nclasses, domain, rep_dim
        self.modalities, self.self_attention, self.importance_loss = modalities, self_attention, importance_loss
        self.data_type, self.encoder_type = data_type, encoder_type
        self.flat = None
        self._sig = None
        self._anchor = None
        self.grad_ready_hook = None
        self._touched_T = 0            # longest stream whose position rows received a gradient since the last exchange
        # nn.TransformerEncoderLayer(d_model, nhead=4) keeps torch's default dropout = 0.1 (prepare_model.py:75); it acts
        # under model.train() (train.py:59) at four sites per layer.  Masks: Philox in the kernels (sais_hip.h), the
        # {seed, offset} state is a device tensor so graph replays draw fresh masks.
        self.dropout_p = 0.1
        self.dropout_seed = 0
        self._rng = None
        self.last_dropout_state = None

    # ------------------------------------------------------------------ engine plumbing
    @property
    def nlayers(self):
        return len(self.transEncoderFrame.layers)

    def _lnames(self, l, enc="transEncoderFrame"):
        return f"{enc}.layers.{l}."

    def _t_names(self):
        out = []
        for l in range(self.nlayers):
            p = self._lnames(l)
            out += [p + "self_attn.in_proj_weight", p + "self_attn.out_proj.weight", p + "linear1.weight",
                    p + "linear2.weight"]
        return out

    def _mil_forward(self, x, f, xpad, fpad):
        """task 'MIL' (prepare_model.py:356-361), inference: per-snippet frame encoder -> relu'd CLS rows + clip position
        rows -> transEncoderClip over the snippets (no mask, no CLS) -> ReLU -> gated-attention MIL head.  Returns the
        reference's (snip_sequence [ns,B,384], snip_reps [B,ns,384], output_logits [B,nclasses], {class: attention [B,ns]}).
        The reference computes the flow stream's clip representations too and drops them (MIL_Head(snip_reps, None)), and it
        indexes the flow tensor unconditionally, so the path exists for modalities 'RGB-Flow' only; training it raises in
        the reference (an in-place add on a ReLU output), so this is an eval() / no-grad path."""
        if self.modalities != 'RGB-Flow':
            raise NotImplementedError("task 'MIL' indexes both streams in the reference (prepare_model.py:358-359): "
                                      "modalities must be 'RGB-Flow'")
        if self.training:
            raise NotImplementedError("task 'MIL' is an inference path: its backward raises inside the reference "
                                      "(getClipReps adds the position rows in place into a ReLU output, :459)")
        if isinstance(x, (list, tuple)):
            raise NotImplementedError("task 'MIL' takes tensor inputs (the reference's list branch never reaches it)")
        x = self._check(x, "x")
        dev = x.device
        xpad = self._mask(xpad, x, dev)
        fl = self._engine(dev)
        B, ns = x.shape[0], x.shape[1]
        S = x.shape[2] + 1
        with torch.no_grad():
            zr, _, _ = self._stream_fwd(x, xpad, save=False, want_attn=False)
            tokens = torch.empty(B * ns, D, dtype=torch.float32, device=dev)
            o = fl.offsets["clip_pos_embeddings.0"]
            assert fl.offsets[f"clip_pos_embeddings.{ns - 1}"] == o + (ns - 1) * D
            ops.mil_forward(zr, S * D, fl.flat[o:o + ns * D], B, ns, tokens)
            nopad = torch.zeros(B, ns, dtype=torch.uint8, device=dev)
            enc, _, _ = self._encoder_fwd(tokens, nopad, B, ns, "transEncoderClip", save=False, want_attn=False)
            reps = torch.empty(B, ns, D, dtype=torch.float32, device=dev)
            logits = torch.empty(B, self.nclasses, dtype=torch.float32, device=dev)
            att = torch.empty(self.nclasses, B, ns, dtype=torch.float32, device=dev)
            cat = lambda fmt: torch.cat([fl.w32(fmt % c).reshape(1, -1) for c in range(3)], 0).contiguous()
            ops.mil_head(enc, B, ns, self.nclasses, fl.w32("attentionA.weight"), fl.w32("attentionA.bias"),
                         fl.w32("attentionB.weight"), fl.w32("attentionB.bias"), cat("attentionModules.%d.weight"),
                         cat("attentionModules.%d.bias").reshape(-1), cat("finalModules.%d.weight"),
                         cat("finalModules.%d.bias").reshape(-1), reps, logits, att)
        return tokens.view(B, ns, D).permute(1, 0, 2), reps, logits, {c: att[c] for c in range(self.nclasses)}

    def _sentinels(self):
        return ["linear.weight", "frame_cls", "frame_pos_embeddings.0", self._lnames(0) + "self_attn.in_proj_weight",
                self._lnames(self.nlayers - 1) + "linear2.weight"]

    def _engine(self, device):
        if self.flat is None or not self.flat.intact() or self.flat.device != device:
            self.flat = FlatParams(self, device, f32_transposes=True)
            self._anchor = torch.zeros(1, device=device, requires_grad=True)
            self._sig = None
        sig = self.flat.signature(self._sentinels())
        if sig != self._sig:
            self.flat.refresh_shadows(self._t_names())
            self._sig = self.flat.signature(self._sentinels())
        return self.flat

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._sig = None
        return r

    def sgd_step(self, lr, grad_scale=1.0):
        self.flat.sgd_step(lr, grad_scale, self._t_names())
        self._sig = self.flat.signature(self._sentinels())

    def dropout_masks(self, state, Bn, S, stream=0):
        """The keep masks a train-mode forward with RNG state `state` (= self.last_dropout_state) applied to stream 0 (RGB) /
        1 (flow) of Bn sequences of S tokens: per layer {attn [Bn,4,S,S], d1 [Bn,S,384], ff [Bn,S,2048], d2 [Bn,S,384]},
        bool.  For tests: the oracle applies the same masks."""
        out, p, dev = [], float(self.dropout_p), state.device
        for l in range(self.nlayers):
            site = (stream * self.nlayers + l) * 4
            m = lambda k, *sh: ops.dropout_mask(int(torch.tensor(sh).prod()), p, state, site + k, dev).view(*sh).bool()
            out.append(dict(attn=m(0, Bn, TH, S, S), d1=m(1, Bn, S, D), ff=m(2, Bn, S, FF), d2=m(3, Bn, S, D)))
        return out

    # ------------------------------------------------------------------ reference signature
    def forward(self, x, f, xlens, flens, task, xpad, fpad, domains=None):
        if task == 'MIL':
            return self._mil_forward(x, f, xpad, fpad)
        if task != 'Prototypes':
            raise NotImplementedError(f"task {task!r}: 'Prototypes' (training and inference) and 'MIL' (inference) are on "
                                      "the MI355X path")
        # multi-domain models ('+' in the domain name): in the two-stream branch the reference sends samples whose domain is
        # not 'NH_02' through linearB (prepare_model.py:405-414); the single-stream branches use linear for everyone
        second = None
        if '+' in self.domain and self.modalities == 'RGB-Flow':
            if domains is None:
                raise ValueError("a multi-domain model needs the per-sample `domains` (prepare_model.py:405-414)")
            ref = x if x is not None else f
            dev = (ref[0] if isinstance(ref, (list, tuple)) else ref).device
            second = torch.tensor([0 if d == 'NH_02' else 1 for d in domains], dtype=torch.uint8, device=dev)
        if isinstance(x, (list, tuple)) or isinstance(f, (list, tuple)):          # TTA versions, :331-346
            n = len(x) if x is not None else len(f)
            if _TTA_MERGE and not self.importance_loss and not (torch.is_grad_enabled() and self.linear.weight.requires_grad):
                return self._forward_tta_merged(x, f, xpad, fpad, second, n)
            embs, attn0, imp0 = [], None, None
            for v in range(n):
                e, a, im = self._forward_one(None if x is None else x[v], None if f is None else f[v],
                                             None if xpad is None else xpad[v], None if fpad is None else fpad[v], second)
                embs.append(e)
                if v == 0:
                    attn0, imp0 = a, im
            return (imp0, embs, attn0) if self.importance_loss else (embs, attn0)
        emb, attn, imp = self._forward_one(x, f, xpad, fpad, second)
        return (imp, emb, attn) if self.importance_loss else (emb, attn)          # :444-448

    def _forward_tta_merged(self, xs, fs, xpads, fpads, second, n):
        """Inference over the TTA versions (prepare_model.py:331-346) as ONE encoder pass: every version of both streams goes
        through the SAME `transEncoderFrame` (:381-399), a sequence's output does not depend on its batch mates, and padded
        positions are masked keys — so the n x (RGB, flow) little passes (6 x 4 layers x 7 launches of 5-15 us for a window
        batch; 20 % of a long video's inference time, LABNOTES R5.3) are stacked along the sequence axis, padded to the longest
        version under the key-padding mask, and run once.  Same values as the per-version passes up to the summation order of
        the masked softmax (tests: test_temporal_tta_list_path, test_tta_merged_pass_equals_per_version_passes)."""
        use_x = self.modalities in ('RGB', 'RGB-Flow')
        use_f = self.modalities in ('Flow', 'RGB-Flow')
        parts = []                                               # (stream, version, [B, ns, T, 384], mask [B * ns, T + 1])
        for name, on, ts, pads in (("x", use_x, xs, xpads), ("f", use_f, fs, fpads)):
            if not on:
                continue
            for v in range(n):
                t = self._check(None if ts is None else ts[v], name)
                parts.append((name, v, t, self._mask(None if pads is None else pads[v], t, t.device)))
        dev = parts[0][2].device
        self._engine(dev)
        Tm = max(t.shape[2] for _, _, t, _ in parts)
        S = Tm + 1
        tot = sum(t.shape[0] * t.shape[1] for _, _, t, _ in parts)
        X = torch.zeros(tot, 1, Tm, D, dtype=torch.float32, device=dev)
        P = torch.ones(tot, S, dtype=torch.uint8, device=dev)
        where, off = {}, 0
        for name, v, t, m in parts:
            k, T = t.shape[0] * t.shape[1], t.shape[2]
            X[off:off + k, 0, :T] = t.reshape(k, T, D)
            P[off:off + k, :T + 1] = m
            where[(name, v)] = (off, k, T)
            off += k
        ref = parts[0][2]
        return self._tta_core(X, P, where, ref.shape[0], ref.shape[1], n, second)

    def _tta_core(self, X, P, where, B, ns, n, second=None):
        """X f32 [tot, 1, Tm, 384] stacked sequences (zero padded), P u8 [tot, Tm + 1] key-padding masks, where[(stream, version)] =
        (first sequence, count, T): one encoder pass, then the head per version.  Returns (embs, version 0's attention map)."""
        use_x = self.modalities in ('RGB', 'RGB-Flow')
        use_f = self.modalities in ('Flow', 'RGB-Flow')
        dev = X.device
        fl = self._engine(dev)
        tot, S = X.shape[0], X.shape[2] + 1
        z, attn, _ = self._stream_fwd(X, P, save=False, want_attn=True)
        z = z.view(tot, S * D)
        embs = []
        for v in range(n):
            zr = z[where[("x", v)][0]:where[("x", v)][0] + B * ns] if use_x else None
            zf = z[where[("f", v)][0]:where[("f", v)][0] + B * ns] if use_f else None
            if use_x and use_f and where[("x", v)][1] != where[("f", v)][1]:
                raise ValueError("RGB and flow streams must have the same batch size and number of snippets")
            rep = torch.empty(B, D, dtype=torch.float32, device=dev)
            emb = torch.empty(B, EMB, dtype=torch.float32, device=dev)
            ops.head_fwd(zr, zf, S * D, B, fl.w32("linear.weight"), fl.w32("linear.bias"), rep, emb, clip_stride_flow=S * D,
                         nsnippets=ns, second=None if second is None else (second, fl.w32("linearB.weight"), fl.w32("linearB.bias")))
            embs.append(emb)
        # the returned map is version 0's, of the RGB stream when there is one (:436-443): its own [T0 + 1, T0 + 1] corner
        o0, k0, T0 = where[("x", 0) if use_x else ("f", 0)]
        attn0 = attn[o0:o0 + k0, :T0 + 1, :T0 + 1].contiguous()
        return embs, attn0

    def _forward_one(self, x, f, xpad, fpad, second=None):
        use_x = self.modalities in ('RGB', 'RGB-Flow')
        use_f = self.modalities in ('Flow', 'RGB-Flow')
        x = self._check(x, "x") if use_x else None
        f = self._check(f, "f") if use_f else None
        dev = (x if x is not None else f).device
        xpad = self._mask(xpad, x, dev) if use_x else None
        fpad = self._mask(fpad, f, dev) if use_f else None
        self._engine(dev)
        if torch.is_grad_enabled() and self.linear.weight.requires_grad:
            emb, attn, imp = _TemporalFn.apply(self, x, f, xpad, fpad, self._anchor, second)
            return emb, attn, (imp if self.importance_loss else None)
        emb, attn, imp, _ = self._forward_kernels(x, f, xpad, fpad, save=False, second=second)
        return emb, attn, imp

    @staticmethod
    def _check(t, name):
        if t is None:
            raise ValueError(f"{name} is required for this modality")
        if not t.is_cuda:
            raise L.SaisHipError("fullModel.forward needs device tensors: the HIP path has no CPU fallback")
        if t.dim() != 4 or t.shape[3] != D:
            raise ValueError(f"{name}: expected [B,nsnippets,T,384], got {tuple(t.shape)}")
        return t.float()

    @staticmethod
    def _mask(pad, t, dev):
        B, ns, T, _ = t.shape
        if pad is None:
            return torch.zeros(B * ns, T + 1, dtype=torch.uint8, device=dev)
        pad = pad.reshape(B * ns, T + 1)                                        # :209
        return pad.to(device=dev, dtype=torch.uint8).contiguous()

    # ------------------------------------------------------------------ kernels
    # All temporal activations are fp32 in HBM (they are tiny: clips*(T+1) rows) and every nn.Linear runs on sais_tgemm
    # (bf16x3 split on the matrix cores, 64 x 64 tiles): the cosine logits inherit ~fp32 accuracy from this half of the
    # path, leaving the whole 1e-3 budget to the bf16 ViT.  The N = 384 GEMMs (out_proj, linear2 and every dX) write RAW
    # split-K slabs; the kernel that consumes them (LayerNorm forward / backward, attention backward, prepare backward)
    # sums the slabs and applies bias / dropout / residual itself, so a layer is 7 launches forward and 8 backward.
    def _raw(self, a, w):
        """a . w^T as raw split-K slabs f32 [nsplit, M, N] (ops.tgemm, TG_RAW)."""
        M, N, K = a.shape[0], w.shape[0], w.shape[1]
        ns = L.load().sais_tgemm_nsplit(M, N, K)          # the split rule lives in the C library
        out = torch.empty(ns, M, N, dtype=torch.float32, device=a.device)
        return ops.tgemm(a, w, L.TG_RAW, out, nsplit=ns)

    def _stream_fwd(self, x, pad, save, want_attn, drop=None, sidx=0):
        fl = self.flat
        dev = x.device
        x = x.reshape(x.shape[0] * x.shape[1], 1, x.shape[2], D)       # (clip, snippet) pairs are independent sequences
        B, _, T, _ = x.shape
        S, M = T + 1, B * (T + 1)
        if T > NPOS:
            raise ValueError("at most 2000 frames (position table, prepare_model.py:67)")
        z = torch.empty(M, D, dtype=torch.float32, device=dev)
        o = fl.offsets["frame_pos_embeddings.0"]
        assert fl.offsets[f"frame_pos_embeddings.{T - 1}"] == o + (T - 1) * D      # rows contiguous in the flat buffer
        ops.temporal_prepare_fwd(x, x.stride(0), x.stride(2), fl.flat[o:o + T * D], fl.w32("frame_cls"), B, T, z, None)
        z, attn, layers = self._encoder_fwd(z, pad, B, S, "transEncoderFrame", save, want_attn, drop, sidx)
        return z, attn, dict(layers=layers, pad=pad, B=B, T=T, x=x, drop=drop, sidx=sidx) if save else None

    def _encoder_fwd(self, z, pad, B, S, enc, save, want_attn, drop=None, sidx=0):
        """The four post-norm layers of `enc` (transEncoderFrame | transEncoderClip) over B sequences of S tokens,
        z f32 [B*S, 384]; pad u8 [B, S] (1 = masked key).  Returns (output, last layer's head-averaged attention, saved)."""
        fl = self.flat
        dev = z.device
        M = B * S
        e32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        layers = []
        attn = None
        for l in range(self.nlayers):
            p = self._lnames(l, enc)
            last = l == self.nlayers - 1
            qkv, ctx = e32(M, 3 * D), e32(M, D)
            if want_attn and last:
                attn = e32(B, S, S)
            if _LAYER_CALLS and ops.TIMER is None:       # one C call per layer (sais_temporal_layer_fwd: the launches below)
                y1, y2 = (e32(M, D), e32(M, D)) if save else (None, None)
                z1, zo, h = e32(M, D), e32(M, D), e32(M, FF)
                m1, r1, m2, r2 = (e32(M), e32(M), e32(M), e32(M)) if save else (None, None, None, None)
                ops.temporal_layer_fwd(self._layer_params(p), B, S, z, pad, qkv, ctx, attn if last else None, y1, z1, m1, r1, h,
                                       y2, zo, m2, r2, drop, (sidx * self.nlayers + l) * 4,
                                       self._ws(L.OP_TEMPORAL_LAYER_FWD, B, S, dev))
                if save:
                    layers.append(dict(z=z, qkv=qkv, ctx=ctx, y1=y1, m1=m1, r1=r1, z1=z1, h=h, y2=y2, m2=m2, r2=r2))
                z = zo
                continue
            ops.tgemm(z, fl.w32(p + "self_attn.in_proj_weight"), L.TG_BIAS, qkv, bias=fl.w32(p + "self_attn.in_proj_bias"))
            # train mode: dropout sites 0-3 of this layer (attention weights, dropout1, dropout, dropout2)
            site = (sidx * self.nlayers + l) * 4
            pd, rng = drop if drop is not None else (0.0, None)
            ops.temporal_attn_fwd(qkv, pad, B, S, ctx, attn if last else None, p_drop=pd, rng=rng, site=site)
            dsite = (lambda k: None if drop is None else (pd, rng, site + k))      # dropout fused into the consumers
            # src = norm1(src + dropout1(out_proj(ctx)))
            y1 = e32(M, D) if save else None
            z1, m1, r1 = e32(M, D), (e32(M) if save else None), (e32(M) if save else None)
            ops.temporal_ln_fwd(self._raw(ctx, fl.w32(p + "self_attn.out_proj.weight")), fl.w32(p + "self_attn.out_proj.bias"),
                                z, fl.w32(p + "norm1.weight"), fl.w32(p + "norm1.bias"), 1e-5, z1, y=y1, mean=m1, rstd=r1,
                                drop=dsite(1))
            # src = norm2(src + dropout2(linear2(dropout(relu(linear1(src))))))
            h = e32(M, FF)
            ops.tgemm(z1, fl.w32(p + "linear1.weight"), L.TG_BIAS_RELU, h, bias=fl.w32(p + "linear1.bias"), drop=dsite(2))
            y2 = e32(M, D) if save else None
            zo, m2, r2 = e32(M, D), (e32(M) if save else None), (e32(M) if save else None)
            ops.temporal_ln_fwd(self._raw(h, fl.w32(p + "linear2.weight")), fl.w32(p + "linear2.bias"), z1,
                                fl.w32(p + "norm2.weight"), fl.w32(p + "norm2.bias"), 1e-5, zo, y=y2, mean=m2, rstd=r2,
                                drop=dsite(3))
            if save:
                layers.append(dict(z=z, qkv=qkv, ctx=ctx, y1=y1, m1=m1, r1=r1, z1=z1, h=h, y2=y2, m2=m2, r2=r2))
            z = zo
        return z, attn, layers

    def _layer_params(self, prefix):
        """ctypes parameter block of one encoder layer (pointers into the flat buffers), rebuilt when those are."""
        f = self.flat
        key = (f.flat.data_ptr(), f.grad.data_ptr(), 0 if f.wt_buf is None else f.wt_buf.data_ptr())
        if getattr(self, "_lp_key", None) != key:
            self._lp_key, self._lp = key, {}
        if prefix not in self._lp:
            self._lp[prefix] = ops.temporal_layer_params(f, prefix)
        return self._lp[prefix]

    def _ws(self, op, B, S, dev, slot=0):
        """Scratch of a layer-level call; `slot` keeps the backward's per-layer workspaces apart (their weight-gradient
        operands are read by ONE deferred launch after the last layer)."""
        if torch.cuda.is_current_stream_capturing():
            return ops.block_workspace(op, B, S, dev)
        key = (op, B, S, str(dev), slot)
        if not hasattr(self, "_wsbuf"):
            self._wsbuf = {}
        if key not in self._wsbuf:
            self._wsbuf[key] = ops.block_workspace(op, B, S, dev)
        return self._wsbuf[key]

    def _prefetch(self, backward):
        """The encoder's ~70 launches per step are a few microseconds each; their weights (35 MB fp32 forward, the same
        again transposed for the backward) were evicted by the ViT's gigabytes since the last step, so every launch would
        pay a first-touch HBM round trip of its own.  One streaming read in front pulls them into the Infinity Cache
        (include/sais_hip.h, sais_touch).  Measured (round 4, two interleaved repetitions on one box): tgemm 11.5 instead of
        12.2 us per launch, the two touch launches cost what that returns (12.89 / 12.99 vs 12.98 / 12.90 ms per step): the
        launches are paced by their own dependent latency chain, not by where the weights sit.  Opt-in: SAIS_TEMPORAL_PREFETCH=1."""
        if not _PREFETCH:
            return
        fl = self.flat
        if backward:
            ops.touch(fl.wt_buf)
        else:
            lo = fl.offsets["transEncoderFrame.layers.0.self_attn.in_proj_weight"]
            hi = fl.offsets["transEncoderClip.layers.0.self_attn.in_proj_weight"]
            ops.touch(fl.flat[lo:hi])

    def _forward_kernels(self, x, f, xpad, fpad, save, second=None):
        fl = self.flat
        self._prefetch(False)
        zr = zf = sr = sf = attn = None
        drop = None
        if self.training and self.dropout_p > 0:
            dev = (x if x is not None else f).device
            if self._rng is None or self._rng.device != dev:
                self._rng = ops.rng_state(self.dropout_seed, dev)
            ops.rng_advance(self._rng)                        # a graph node: every replay draws fresh masks
            self.last_dropout_state = self._rng.clone()       # what this forward and its backward regenerate the masks from
            drop = (float(self.dropout_p), self.last_dropout_state)
        if x is not None:
            zr, attn, sr = self._stream_fwd(x, xpad, save, want_attn=True, drop=drop, sidx=0)
        if f is not None:
            zf, fattn, sf = self._stream_fwd(f, fpad, save, want_attn=(x is None), drop=drop, sidx=1)
            if x is None:
                attn = fattn
        ref = x if x is not None else f
        B, ns = ref.shape[0], ref.shape[1]
        # the two streams may have different lengths (inference: 15 RGB frames vs 1-2 flow frames per window)
        Sx = x.shape[2] + 1 if x is not None else f.shape[2] + 1
        Sf = f.shape[2] + 1 if f is not None else Sx
        if x is not None and f is not None and tuple(x.shape[:2]) != tuple(f.shape[:2]):
            raise ValueError("RGB and flow streams must have the same batch size and number of snippets")
        rep = torch.empty(B, D, dtype=torch.float32, device=ref.device)
        emb = torch.empty(B, EMB, dtype=torch.float32, device=ref.device)
        ops.head_fwd(zr, zf, (Sx if zr is not None else Sf) * D, B, fl.w32("linear.weight"), fl.w32("linear.bias"), rep, emb,
                     clip_stride_flow=Sf * D, nsnippets=ns,
                     second=None if second is None else (second, fl.w32("linearB.weight"), fl.w32("linearB.bias")))
        imp = None
        if self.importance_loss:                              # importance_function(full RGB sequence), :419-421
            imp = torch.empty(B, ns, Sx, 1, dtype=torch.float32, device=ref.device)
            ops.importance_fwd(zr, fl.w32("importance_function.weight"), fl.w32("importance_function.bias"), B * ns * Sx, imp)
        saved = dict(sr=sr, sf=sf, zr=zr, zf=zf, rep=rep, B=B, ns=ns, Sx=Sx, Sf=Sf, second=second,
                     xshape=None if x is None else x.shape, fshape=None if f is None else f.shape) if save else None
        return emb, attn, imp, saved

    def _stream_bwd(self, s, dz, need_dx):
        """dz: f32 [M,384] gradient wrt the stream's final (pre-ReLU) encoder output."""
        fl = self.flat
        dev = dz.device
        B, T = s["B"], s["T"]
        S, M = T + 1, B * (T + 1)
        e32 = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)
        slabs, add = None, dz                      # the gradient entering a layer = sum of `slabs` (raw dX GEMM output) + add
        layer_calls = _LAYER_CALLS and ops.TIMER is None
        dw_items = ops.tn_items(4 * self.nlayers) if layer_calls and _DW_DEFER else None
        keep = []                                  # per-layer gradients the deferred launch still reads
        for l in reversed(range(self.nlayers)):
            p = self._lnames(l)
            a = s["layers"][l]
            # dropout backward = the same mask on the branch gradient (the residual path keeps the un-dropped one), fused:
            # the LayerNorm backward emits dropout(dx) as a second output, and the drelu epilogue applies the FFN mask
            # (a["h"] is the DROPPED relu output, so dropped units are already zero there: the mask rescales the kept ones)
            drop = s.get("drop")
            site = (s.get("sidx", 0) * self.nlayers + l) * 4
            pd, rng = drop if drop is not None else (0.0, None)
            dsite = (lambda k: None if drop is None else (pd, rng, site + k))
            if layer_calls:                              # one C call per layer (sais_temporal_layer_bwd)
                ns = L.load().sais_tgemm_nsplit(M, D, 3 * D)
                dx_slabs, dy1 = e32(ns, M, D), e32(M, D)
                ws = self._ws(L.OP_TEMPORAL_LAYER_BWD, B, S, dev, slot=l)
                ops.temporal_layer_bwd(self._layer_params(p), B, S, a, s["pad"], slabs, add, dx_slabs, dy1, drop, site, ws,
                                       dw_items=None if dw_items is None else (dw_items, 4 * l))
                keep += [ws, dy1]
                slabs, add = dx_slabs, dy1
                continue
            dy2 = e32(M, D)
            dt2 = dy2 if drop is None else e32(M, D)
            ops.temporal_ln_bwd(slabs, add, a["y2"], a["m2"], a["r2"], fl.w32(p + "norm2.weight"), dy2,
                                dx_drop=None if drop is None else dt2, drop=dsite(3),
                                dgamma=fl.g(p + "norm2.weight"), dbeta=fl.g(p + "norm2.bias"))
            dh = e32(M, FF)
            ops.tgemm(dt2, fl.wt16[p + "linear2.weight"], L.TG_DRELU, dh, aux=a["h"], drop=dsite(2))
            dy1 = e32(M, D)                                   # LayerNorm'(dy2 (residual) + dh . W1)
            dt1 = dy1 if drop is None else e32(M, D)
            ops.temporal_ln_bwd(self._raw(dh, fl.wt16[p + "linear1.weight"]), dy2, a["y1"], a["m1"], a["r1"],
                                fl.w32(p + "norm1.weight"), dy1, dx_drop=None if drop is None else dt1, drop=dsite(1),
                                dgamma=fl.g(p + "norm1.weight"), dbeta=fl.g(p + "norm1.bias"))
            dqkv = e32(M, 3 * D)
            ops.temporal_attn_bwd(a["qkv"], s["pad"], B, S, self._raw(dt1, fl.wt16[p + "self_attn.out_proj.weight"]), dqkv,
                                  p_drop=pd, rng=rng, site=site)
            # the four weight / bias gradients of the layer in one launch (M is a few hundred rows: launch-bound); one M-split:
            # every workgroup owns its output tile and accumulates without atomics
            ops.gemm_tn_grouped([
                (dt2, a["h"], fl.g(p + "linear2.weight"), fl.g(p + "linear2.bias")),
                (dh, a["z1"], fl.g(p + "linear1.weight"), fl.g(p + "linear1.bias")),
                (dt1, a["ctx"], fl.g(p + "self_attn.out_proj.weight"), fl.g(p + "self_attn.out_proj.bias")),
                (dqkv, a["z"], fl.g(p + "self_attn.in_proj_weight"), fl.g(p + "self_attn.in_proj_bias"))], M, nsplit=1)
            # gradient wrt the layer input = dy1 (residual) + dqkv . Win: left as slabs for the next consumer
            slabs, add = self._raw(dqkv, fl.wt16[p + "self_attn.in_proj_weight"]), dy1
        if dw_items is not None:       # the 4 x nlayers weight / bias gradient GEMMs of the encoder in ONE launch (they are off the dX chain)
            ops.temporal_dw_deferred(dw_items, 4 * self.nlayers, M)
        del keep
        x = s["x"]
        dx = torch.empty_like(x) if need_dx else None
        self._touched_T = max(self._touched_T, T)
        o = fl.offsets["frame_pos_embeddings.0"]
        ops.temporal_prepare_bwd(add, slabs, B, T, dx, 0 if dx is None else dx.stride(0), 0 if dx is None else dx.stride(2),
                                 False, fl.grad[o:o + T * D], fl.g("frame_cls"))
        return dx

    def _backward_kernels(self, saved, demb, needs, dimp=None):
        fl = self.flat
        fl.attach_grads()
        self._prefetch(True)
        B, ns, Sx, Sf = saved["B"], saved["ns"], saved["Sx"], saved["Sf"]
        zr, zf = saved["zr"], saved["zf"]
        dzr = torch.zeros_like(zr) if zr is not None else None
        dzf = torch.zeros_like(zf) if zf is not None else None
        sec = saved.get("second")
        ops.head_bwd(demb, fl.w32("linear.weight"), saved["rep"], zr, zf, (Sx if zr is not None else Sf) * D, B,
                     fl.g("linear.weight"), fl.g("linear.bias"), dzr, dzf, clip_stride_flow=Sf * D, nsnippets=ns,
                     second=None if sec is None else (sec, fl.w32("linearB.weight"), fl.g("linearB.weight"),
                                                      fl.g("linearB.bias")))
        if dimp is not None:
            ops.importance_bwd(dimp, zr, fl.w32("importance_function.weight"), B * ns * Sx, dzr,
                               fl.g("importance_function.weight"), fl.g("importance_function.bias"))
        dx = self._stream_bwd(saved["sr"], dzr, needs[0]) if zr is not None else None
        df = self._stream_bwd(saved["sf"], dzf, needs[1]) if zf is not None else None
        if self.grad_ready_hook:
            self.grad_ready_hook(0, fl.numel)
        if dx is not None:
            dx = dx.view(saved["xshape"])
        if df is not None:
            df = df.view(saved["fshape"])
        return dx, df
