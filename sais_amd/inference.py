"""Inference surface of SAIS on MI355X: frozen ViT feature extraction (fixed-shape batches replayed from a
hipGraph) -> Custom_Gestures sliding windows with 3 test-time-augmentation index sets -> temporal encoder ->
embeddings + attention maps, written in the reference's output formats.

Reference behaviour restated here (paths relative to SAIS/scripts/):
  extract_representations.extractFeatures :351-378   eval / no_grad loop over frame batches -> [nframes,384]
  prepare_dataset.py:1705-1727   windows: duration 0.5 s x 30 fps = 15 frames, hop 15:
                                 nsamples = (total - 15)//15 + 1, StartFrame = 15 n, EndFrame = StartFrame + 15
  prepare_dataset.py:2642-2666   startIdx = StartFrame-1, endIdx = EndFrame-1, jump = (end-start)//10 = 1;
                                 TTA index sets arange(start+{0,3,6}, end) -> 15 / 12 / 9 frames; the first window
                                 starts at -1, which numpy wraps to the LAST frame (App. B.6, reproduced);
                                 flow rows = unique(idx // 15) kept if < len(flow_reps)  (so -1 -> last row too)
  prepare_dataset.py:2839-2899   pad_collate, tuple branch: per TTA version zero-pad to the batch max and build the
                                 bool key-padding mask [B,1,maxT+1]
  perform_training.py:96-185, train.py:113-119   per-sample reps lists (tuple of 3 under TTA), attention list of
                                 per-batch [B,T+1,T+1], labels, videonames, logits=[]
"""
import os

import numpy as np
import torch

from . import _lib as L

DURATION_FRAMES, HOP_FRAMES, FLOW_JUMP = 15, 15, 15
TTA_OFFSETS = (0, 3, 6)


# --------------------------------------------------------------------------- window sampler (host logic)
def gesture_windows(total_frames):
    """[(StartFrame, EndFrame)] — prepare_dataset.py:1716-1719."""
    nsamples = (total_frames - DURATION_FRAMES) // HOP_FRAMES + 1
    return [(n * HOP_FRAMES, n * HOP_FRAMES + DURATION_FRAMES) for n in range(max(nsamples, 0))]


def tta_indices(start_frame, end_frame):
    """Three RGB index sets of one window — prepare_dataset.py:2642-2651 (may contain -1)."""
    s, e = start_frame - 1, end_frame - 1
    jump = (e - s) // 10
    return [list(np.arange(s + off, e, jump)) for off in TTA_OFFSETS]


def flow_rows(indices, nflow):
    """prepare_dataset.py:2660-2666: unique(idx // 15) with idx < len(flow_reps) (negative rows survive the filter)."""
    rows = np.unique([i // FLOW_JUMP for i in indices])
    return [int(r) for r in rows if r < nflow]


def sample_window(rgb_reps, flow_reps, start_frame, end_frame):
    """One dataset item: ((x0,x1,x2), (f0,f1,f2)) with x_v [1,T_v,384], f_v [1,Tf_v,384]  (:2653-2700)."""
    xs, fs = [], []
    for idx in tta_indices(start_frame, end_frame):
        xs.append(rgb_reps[idx].unsqueeze(0))                       # negative index wraps, as numpy does there
        fs.append(flow_reps[flow_rows(idx, flow_reps.shape[0])].unsqueeze(0))
    return tuple(xs), tuple(fs)


def pad_collate_tta(items):
    """pad_collate, Prototypes / tuple branch (:2841-2873).  items: list of ((x0,x1,x2),(f0,f1,f2)).
    Returns per version: padded x [B,1,maxT,384], mask [B,1,maxT+1] (True = masked), lens — and the same for flow."""
    out = {"x": [], "xpad": [], "xlens": [], "f": [], "fpad": [], "flens": []}
    nver = len(items[0][0])
    for v in range(nver):
        for key, which in (("x", 0), ("f", 1)):
            seqs = [it[which][v] for it in items]                    # each [1,T,384]
            lens = [s.shape[1] for s in seqs]
            maxT = max(lens)
            B = len(seqs)
            padded = seqs[0].new_zeros(B, 1, maxT, seqs[0].shape[-1])
            mask = torch.zeros(B, 1, maxT + 1, dtype=torch.bool, device=seqs[0].device)
            for b, s in enumerate(seqs):
                padded[b, :, :lens[b]] = s
                mask[b, :, lens[b] + 1:] = True                       # createPaddingMask :2798-2806
            out[key].append(padded)
            out[key + "pad"].append(mask)
            out[key + "lens"].append(lens)
    return out


def collate_windows_tta(rgb_reps, flow_reps, wins, pad_flow_to=0):
    """sample_window + pad_collate_tta over MANY windows of one video at once: the same dict of tensors as
    pad_collate_tta([sample_window(rgb_reps, flow_reps, s, e) for s, e in wins]) (held to it in tests/test_host_cpu.py), built
    with ONE gather per TTA version and stream instead of six small ones per window — the index arithmetic
    (prepare_dataset.py:2642-2666: wrap-around -1, flow rows unique(idx // 15) < len(flow_reps)) runs on the host in numpy."""
    dev = rgb_reps.device
    nflow = flow_reps.shape[0]
    B = len(wins)
    out = {"x": [], "xpad": [], "xlens": [], "f": [], "fpad": [], "flens": []}
    starts = np.asarray([s for s, _ in wins], dtype=np.int64)
    uniform = all(e - s == DURATION_FRAMES for s, e in wins)     # what gesture_windows produces: jump = 15 // 10 = 1
    per_win = None if uniform else [tta_indices(s, e) for s, e in wins]
    for v, off in enumerate(TTA_OFFSETS):
        if uniform:
            # closed form of tta_indices / flow_rows for 15-frame windows: indices s - 1 + off .. e - 2, their flow rows the one or
            # two values of idx // 15 (floor division: -1 // 15 = -1 survives the `< nflow` filter, as in the reference)
            L = DURATION_FRAMES - off
            first = starts - 1 + off
            rix = first[:, None] + np.arange(L, dtype=np.int64)[None, :]
            rkeep = np.ones((B, L), dtype=bool)
            r0, r1 = first // FLOW_JUMP, (first + L - 1) // FLOW_JUMP
            fix = np.stack([r0, r1], 1)
            fkeep = np.stack([r0 < nflow, (r1 != r0) & (r1 < nflow)], 1)
            fix = np.where(fkeep, fix, 0)
            both = ((rix, rkeep), (fix, fkeep))
        else:
            rgb_idx = [per_win[b][v] for b in range(B)]
            both = []
            for idx in (rgb_idx, [flow_rows(ix, nflow) for ix in rgb_idx]):
                m = max(max(len(r) for r in idx), 1)
                ix, keep = np.zeros((B, m), dtype=np.int64), np.zeros((B, m), dtype=bool)
                for b, row in enumerate(idx):
                    ix[b, :len(row)] = row
                    keep[b, :len(row)] = True
                both.append((ix, keep))
        for key, reps, (ix, keep) in (("x", rgb_reps, both[0]), ("f", flow_reps, both[1])):
            lens = keep.sum(1)
            maxT = int(lens.max())
            if key == "f" and pad_flow_to:                   # static shapes for graph replay: padding is masked, values unchanged
                maxT = max(maxT, pad_flow_to)
            if ix.shape[1] < max(maxT, 1):
                padc = max(maxT, 1) - ix.shape[1]
                ix, keep = np.pad(ix, ((0, 0), (0, padc))), np.pad(keep, ((0, 0), (0, padc)))
            g = reps[torch.from_numpy(ix).to(dev)]                        # negative indices wrap, as numpy does in the reference
            g = (g * torch.from_numpy(keep).to(dev).unsqueeze(-1).to(g.dtype))[:, :maxT]
            mask = torch.from_numpy(np.arange(maxT + 1)[None, :] > lens[:, None]).unsqueeze(1)      # createPaddingMask :2798-2806
            out[key].append(g.unsqueeze(1).contiguous())
            out[key + "pad"].append(mask.to(dev))
            out[key + "lens"].append([int(n) for n in lens])
    return out


# --------------------------------------------------------------------------- ViT feature extraction
class FeatureExtractor:
    """extractFeatures (:351-378) with the fixed-shape ViT forward captured once into a hipGraph and replayed per
    batch (the launch-bound part of inference: ~90 kernel launches per batch collapse into one graph launch)."""

    MAX_GRAPHS = 16                                                   # captured shapes kept per extractor (least recently used out)

    def __init__(self, vit, batch_size=32, use_graph=True, tail_batch=None, tail_round=2):
        """batch_size frames per replay; tail_batch: the shape used for what is left at the end of a video (and for short
        inputs such as the flow maps: 34 of them for a 512-frame video), so that a large main batch — the GEMM kernels are most
        efficient from ~50 k token rows = 256 frames on — does not turn the remainder into a mostly-padding replay.
          None    the remainder is padded to batch_size (one captured shape);
          int     a second, fixed captured shape (< batch_size); what is left is padded to it;
          "fit"   the remainder itself, rounded up to a multiple of tail_round: one captured shape per distinct remainder (at
                  most MAX_GRAPHS kept).  34 flow maps then cost 34 frames of work instead of 64: 47.5 k -> 50.3 k frames/s on
                  the 512-frame video (LABNOTES R6.6)."""
        self.vit = vit.eval()
        self.bs = batch_size
        self.fit = tail_batch == "fit"
        self.tail_round = max(1, int(tail_round))
        self.tail = tail_batch if (not self.fit and tail_batch and tail_batch < batch_size) else None
        self.use_graph = use_graph
        self._graphs = {}                                             # frames per replay -> (graph, static_in, static_out)
        self._sig = None

    def shape_for(self, left):
        """Frames per forward pass for `left` remaining frames: the padded shape, the same for replayed and eager passes."""
        if left >= self.bs:
            return self.bs
        if self.fit:
            return min(self.bs, -(-left // self.tail_round) * self.tail_round)
        return self.bs if self.tail is None else self.tail

    def _capture(self, device, bs):
        static_in = torch.zeros(bs, 3, 224, 224, device=device)
        with torch.no_grad():
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2):                                    # warm-up: allocator + LDS attributes + shadows
                    self.vit(static_in)
            torch.cuda.current_stream().wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = self.vit(static_in)
        while len(self._graphs) >= self.MAX_GRAPHS:                   # dicts keep insertion order: the first key is the oldest use
            old = next(k for k in self._graphs if k != self.bs)
            del self._graphs[old]
        self._graphs[bs] = (graph, static_in, static_out)
        self._sig = self._weights_sig(device)

    def _weights_sig(self, device):
        """The captured launches hold pointers into the ViT's weight shadows and were recorded against one set of weights: a later
        load_state_dict / optimizer step (FlatParams.signature) or a reallocated flat buffer makes every captured graph stale."""
        fl = self.vit._engine(device)
        return (fl.signature(self.vit._sentinels), fl.flat.data_ptr())

    # the attributes older callers / tests read: the main shape's graph
    @property
    def graph(self):
        return self._graphs.get(self.bs, (None,))[0]

    @torch.no_grad()
    def __call__(self, frames):
        """frames [N,3,224,224] (device) -> [N,384] fp32."""
        if not frames.is_cuda:
            raise L.SaisHipError("FeatureExtractor needs device tensors: the HIP path has no CPU fallback")
        N = frames.shape[0]
        out = torch.empty(N, 384, device=frames.device)
        i = 0
        while i < N:
            left = N - i
            bs = self.shape_for(left)
            n = min(bs, left)
            if self.use_graph:
                if self._graphs and self._weights_sig(frames.device) != self._sig:
                    self._graphs.clear()                              # the weights changed since the capture: never replay those
                if bs not in self._graphs:
                    self._capture(frames.device, bs)
                graph, static_in, static_out = self._graphs[bs] = self._graphs.pop(bs)      # re-insert: most recently used last
                static_in[:n].copy_(frames[i:i + n])
                if n < bs:
                    static_in[n:].zero_()
                graph.replay()
                out[i:i + n].copy_(static_out[:n])
            else:                                                     # the SAME padded shapes as the captured graphs, so that
                x = frames[i:i + n].float()                           # eager and replayed features are bit-identical (the
                if n < bs:                                            # dispatch regime of the kernels depends on the row count)
                    x = torch.cat([x, x.new_zeros(bs - n, 3, 224, 224)])
                out[i:i + n] = self.vit(x)[:n]
            i += n
        return out


def collate_windows_merged(rgb_reps, flow_reps, wins):
    """The 15-frame windows of gesture_windows (hop 15, jump 1) of ONE video as a single stacked batch for fullModel._tta_core:
    X f32 [6 B, 1, 15, 384] (RGB versions 0-2, then flow versions 0-2; zero padded), P u8 [6 B, 16] (1 = masked key) and the
    layout dict.  Same rows, zeros and masks as collate_windows_tta / pad_collate_tta give version by version
    (tests/test_host_cpu.py), built from ONE index array, one host-to-device copy and one gather over [rgb_reps; flow_reps]."""
    dev = rgb_reps.device
    N, nflow = rgb_reps.shape[0], flow_reps.shape[0]
    B = len(wins)
    starts = np.asarray([s for s, _ in wins], dtype=np.int64)
    assert all(e - s == DURATION_FRAMES for s, e in wins)
    Tm = DURATION_FRAMES
    idx = np.zeros((6 * B, Tm), dtype=np.int64)
    lens = np.zeros(6 * B, dtype=np.int64)
    where = {}
    for v, off in enumerate(TTA_OFFSETS):
        Lv = DURATION_FRAMES - off
        first = starts - 1 + off                                     # may be -1: wraps to the last frame, as in the reference
        rix = first[:, None] + np.arange(Lv, dtype=np.int64)[None, :]
        idx[v * B:(v + 1) * B, :Lv] = np.where(rix < 0, rix + N, rix)
        lens[v * B:(v + 1) * B] = Lv
        where[("x", v)] = (v * B, B, Lv)
        r0, r1 = first // FLOW_JUMP, (first + Lv - 1) // FLOW_JUMP    # floor division: -1 // 15 = -1 survives `< nflow` and wraps
        k0, k1 = r0 < nflow, (r1 != r0) & (r1 < nflow)
        # kept rows first (unique() order), as flow_reps[rows] gives them
        a = np.where(k0, r0, r1)
        b = np.where(k0 & k1, r1, 0)
        n_f = k0.astype(np.int64) + k1.astype(np.int64)
        fo = (3 + v) * B
        idx[fo:fo + B, 0] = N + np.where(a < 0, a + nflow, a) * (n_f > 0)
        idx[fo:fo + B, 1] = N + np.where(b < 0, b + nflow, b) * (n_f > 1)
        lens[fo:fo + B] = n_f
        where[("f", v)] = (fo, B, 2)
    keep = np.arange(Tm)[None, :] < lens[:, None]
    idx = np.where(keep, idx, 0)
    packed = torch.from_numpy(np.concatenate([idx, keep.astype(np.int64)], 1)).to(dev, non_blocking=True)
    table = torch.cat([rgb_reps, flow_reps], 0)
    X = (table[packed[:, :Tm]] * packed[:, Tm:].unsqueeze(-1).to(table.dtype)).unsqueeze(1)
    P = torch.cat([torch.zeros(6 * B, 1, dtype=torch.uint8, device=dev), (1 - packed[:, Tm:]).to(torch.uint8)], 1)
    return X, P, where


# --------------------------------------------------------------------------- windowed temporal inference
class _WindowGraph:
    """The temporal encoder over ONE fixed-size chunk of windows (all three TTA versions, both streams) captured into a
    hipGraph: static inputs x_v [chunk,1,T_v,384], f_v [chunk,1,FLOW_PAD,384] and their key-padding masks; static outputs.  A
    chunk is ~170 launches of 5-15 us kernels issued from Python; replayed it is one graph launch.  Shapes depend only on the chunk
    size, so ONE capture per model serves every video.  OPT-IN (`run_windows(use_graph=True)`) and not used by the CLI or the
    benchmark: measured SLOWER on a 512-frame video (5.6 vs 3.4 ms for the windows half, LABNOTES R5.7) — the ~170 tiny kernels
    are device-bound at ~8 us each, so removing the host's launch work buys nothing, while fixed chunks compute 64 windows for 34."""
    FLOW_PAD = 2                                             # unique(idx // 15) of <= 15 consecutive frames spans <= 2 flow rows

    def __init__(self, model, c, use_f):
        self.model, self.use_f = model, use_f
        self.x = [t.clone() for t in c["x"]]
        self.f = [t.clone() for t in c["f"]]
        self.xpad = [t.clone() for t in c["xpad"]]
        self.fpad = [t.clone() for t in c["fpad"]]
        self.xlens, self.flens = c["xlens"], c["flens"]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):                               # warm-up: workspaces, engines, allocator
                self._call()
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self._call()

    def _call(self):
        return self.model(self.x, self.f if self.use_f else None, self.xlens, self.flens, 'Prototypes', self.xpad,
                          self.fpad if self.use_f else None, None)

    def replay(self, c):
        for dst, src in zip(self.x + self.f + self.xpad + self.fpad, c["x"] + c["f"] + c["xpad"] + c["fpad"]):
            dst.copy_(src)
        self.graph.replay()
        return self.out


@torch.no_grad()
def run_windows(model, rgb_reps, flow_reps, videoname="video", batch_size=2, total_frames=None, rank=0, world_size=1,
                compute_batch=256, use_graph=False):
    """The `Custom_inference` phase of single_epoch (perform_training.py:71-185) over one video.
    Returns the dict train.py:116 saves as reps_and_labels_<phase>, the attention list (:117) and the importance list
    (:118; per-batch [B,1,T+1,1] tensors with `-il`, else empty).  `total_frames` = the video's row count in
    paths/Custom_Paths.csv, which is what sizes the windows in the reference (prepare_dataset.py:1705-1727); default:
    the number of feature rows.
    world_size > 1: the batches of `batch_size` windows are the reference's own (same composition, same
    padding), rank r runs a contiguous range of them and every rank returns the lists merged in batch order.
    `batch_size` (main.sh: -bs 2) fixes the SHAPE of the outputs (one attention tensor per batch of that many windows, as
    train.py:117 saves them); the arithmetic runs over up to `compute_batch` windows per call of the model — a window's
    outputs do not depend on its batch mates (padding is masked, rows are independent), and 17 x 3 x 2 little forward passes
    with a host round trip each were 63 % of a 512-frame video's inference time (LABNOTES R5.3).  Embeddings, attention
    and importances leave the device ONCE at the end.
    use_graph=True: the windows are processed in fixed chunks of min(compute_batch, 32) (the last chunk padded with repeats
    of its last window, flow rows padded to two under the mask) and each chunk is one replay of a hipGraph captured once per
    model (`_WindowGraph`); same values up to the masked padding's summation order."""
    from .parallel import gather_in_rank_order, shard_range
    model.eval()
    wins = gesture_windows(rgb_reps.shape[0] if total_frames is None else total_frames)
    reps = ([], [], [])
    attention, labels, names, importance = [], [], [], []
    starts = list(range(0, len(wins), batch_size))
    lo, hi = shard_range(len(starts), rank, world_size)
    mine = wins[starts[lo]:starts[hi - 1] + batch_size] if hi > lo else []
    use_f = model.modalities in ("Flow", "RGB-Flow")
    cb = max(batch_size, (int(compute_batch) // batch_size) * batch_size)     # whole batches per call
    if use_graph:
        cb = min(cb, 32)
    if model.modalities == "Flow":
        # the returned map is the FLOW stream's [B, Tf + 1, Tf + 1] with Tf the longest flow sequence of the BATCH (one or two rows):
        # a wider compute chunk would change the shape the reference saves for a batch whose windows all have one flow row
        cb, use_graph = batch_size, False
    emb_parts, attn_parts, imp_parts = ([], [], []), [], []
    from . import temporal as _tmod
    fast = (not use_graph and _tmod._TTA_MERGE and hasattr(model, "_tta_core") and model.modalities == "RGB-Flow"
            and not model.importance_loss and '+' not in getattr(model, "domain", "")
            and all(e - s == DURATION_FRAMES for s, e in mine))
    for i in range(0, len(mine), cb):
        chunk = mine[i:i + cb]
        n = len(chunk)
        if use_graph:
            c = collate_windows_tta(rgb_reps, flow_reps, chunk + [chunk[-1]] * (cb - n), pad_flow_to=_WindowGraph.FLOW_PAD)
            graphs = model.__dict__.setdefault("_window_graphs", {})
            # the captured launches hold pointers into the model's weight shadows: a graph is valid only for the weights it was captured
            # with (FlatParams.signature changes on load_state_dict / an optimizer step, and shadows may then be reallocated)
            sig = model._engine(rgb_reps.device).signature(model._sentinels())
            key = (cb, str(rgb_reps.device), tuple(tuple(t.shape) for t in c["x"] + c["f"]), sig, model.flat.flat.data_ptr())
            for k in [k for k in graphs if k[:3] == key[:3] and k != key]:
                del graphs[k]                                    # stale weights: drop, never replay
            if key not in graphs:
                graphs[key] = _WindowGraph(model, c, use_f)
            out = graphs[key].replay(c)
        elif fast:
            # one stacked batch, one encoder pass (fullModel._tta_core): no per-version tensors on the way in
            X, P, where = collate_windows_merged(rgb_reps, flow_reps, chunk)
            out = model._tta_core(X, P, where, n, 1, 3)
        else:
            c = collate_windows_tta(rgb_reps, flow_reps, chunk)
            out = model(c["x"], c["f"] if use_f else None, c["xlens"], c["flens"], 'Prototypes', c["xpad"],
                        c["fpad"] if use_f else None, None)
        if model.importance_loss:                                    # (importances, embs, attn), prepare_model.py:445-446
            imp, embs, attn = out
            imp_parts.append(imp.detach()[:n].clone())               # perform_training.py:139-141
        else:
            embs, attn = out
        for v in range(3):
            e = embs[v].detach()[:n]
            emb_parts[v].append(e.clone() if use_graph else e)        # clone: the graph's static outputs are overwritten by the next replay
        attn_parts.append(attn.detach()[:n].clone() if use_graph else attn.detach()[:n])
    if mine:
        # ONE device -> host copy: [embeddings of the three versions | attention maps] packed row-wise per window; the per-sample /
        # per-batch tensors the reference saves are views of that one host tensor (torch.save stores the shared storage once)
        embs_d = [torch.cat(p) if len(p) > 1 else p[0] for p in emb_parts]
        attn_d = torch.cat(attn_parts) if len(attn_parts) > 1 else attn_parts[0]
        nw, E = attn_d.shape[0], embs_d[0].shape[1]
        ashape = attn_d.shape[1:]
        packed = torch.cat(embs_d + [attn_d.reshape(nw, -1)], 1).cpu()
        imp_cpu = torch.cat(imp_parts).cpu() if imp_parts else None
        for v in range(3):
            reps[v].extend(packed[:, v * E:(v + 1) * E].unbind(0))
        attn_cpu = packed[:, 3 * E:].reshape(nw, *ashape)
        for i in range(0, len(mine), batch_size):                    # the reference's per-batch tensors
            attention.append(attn_cpu[i:i + batch_size])
            if imp_cpu is not None:
                importance.append(imp_cpu[i:i + batch_size].clone())
        labels += [torch.tensor(0, dtype=torch.long)] * len(mine)            # placeholder label (:2637)
        names += [videoname] * len(mine)
    if world_size > 1:
        parts = gather_in_rank_order((reps, attention, labels, names, importance), world_size)
        reps = tuple([e for p in parts for e in p[0][v]] for v in range(3))
        attention = [a for p in parts for a in p[1]]
        labels = [l for p in parts for l in p[2]]
        names = [n for p in parts for n in p[3]]
        importance = [m for p in parts for m in p[4]]
    return {"reps": reps, "labels": labels, "videonames": names, "logits": []}, attention, importance


def save_inference_outputs(savepath, phase, reps_and_labels, attention, importance=()):
    """train.py:113-119 — rank 0 writes reps_and_labels_<ph>, attention_<ph>, importance_<ph> with torch.save."""
    os.makedirs(savepath, exist_ok=True)
    torch.save(reps_and_labels, os.path.join(savepath, f"reps_and_labels_{phase}"))
    torch.save(attention, os.path.join(savepath, f"attention_{phase}"))
    torch.save(list(importance), os.path.join(savepath, f"importance_{phase}"))


def tta_probs(reps_and_labels, prototypes):
    """process_inference_results.py:76-91,218: calcProbs per TTA version, then the mean over versions."""
    from .loss import cosine_logits_and_probs
    dev = next(iter(prototypes.values())).device
    probs = []
    for v in range(3):
        emb = torch.stack(reps_and_labels["reps"][v]).to(dev)
        probs.append(cosine_logits_and_probs(emb, prototypes)[1])
    return torch.stack(probs).mean(0)
