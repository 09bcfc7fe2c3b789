"""Data-parallel gradient exchange for the SAIS hot path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

Clips are independent, so the only exchange step is the gradient all-reduce.  Gradients live in flat
buffers (sais_amd/flat.py); the backward pass calls `grad_ready_hook(lo, hi)` as soon as a contiguous
slice is final (per ViT block, last block first), and the slice is all-reduced asynchronously:
ProcessGroupNCCL enqueues it on its own stream behind the work already queued on the compute stream, so
it overlaps the rest of backward.  `wait()` joins right before the SGD step, which applies the 1/world
average through its `grad_scale` argument.  The reference's (disabled) DDP needed
find_unused_parameters=True (prepare_model.py:549): here only the touched slices are exchanged.
"""
import torch
import torch.distributed as dist


class GradSync:
    """Bucketed exchange (8-16 MB buckets in reverse-layer order; the reference's disabled DDP,
    prepare_model.py:546-553, would have used DDP's 25 MB default).  Slices arrive from the backward hooks in a fixed order
    (temporal slices, final norm, ViT blocks last..first, embedding); a slice that is ADJACENT in memory to the open bucket
    extends it (the ViT blocks are consecutive ranges of one flat buffer, so three of them form one 21 MB all-reduce), a
    bucket is issued as soon as it holds >= `bucket_bytes`, and slices below `small_bytes` (head, CLS, position rows,
    prototypes, final norm, embedding) are packed into ONE staging tensor that is exchanged at wait() and scattered back.
    xGMI rings are per-link bound: 6 collectives of 2-35 MB per step instead of 20 of ~7 MB (config 2).  Bucket boundaries
    depend only on the slice sizes and the hook order, which are the same on every rank."""

    def __init__(self, world=None, active=None, bucket_bytes=16 << 20, small_bytes=2 << 20, payload_dtype=None):
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.active = self.world > 1 if active is None else active     # active with world 1: exercises the path
        self.bucket_bytes, self.small_bytes = int(bucket_bytes), int(small_bytes)
        self.pending = []
        self.bytes = 0                      # bytes handed to all_reduce since the last wait()
        self._temporal = None               # fullModel whose touched slices still have to be exchanged this step
        self._open = None                   # [base, lo, hi, nslices]: the bucket being filled (element range of `base`)
        self._small = []                    # small slices waiting to be packed
        self._staged = []                   # (staging tensor, slices) in flight: packed small slices, reduced-precision buckets
        # what travels over xGMI: None / torch.float32 = the fp32 gradients themselves, in place; torch.bfloat16 = a bf16 copy
        # of every bucket (half the bytes: 61 instead of 122 MB per step at config 2), summed by RCCL in bf16 and written
        # back as fp32 at wait().  The rings are per-link bound, so bytes are what scaling efficiency pays for; the rounding
        # (2^-9 relative per addend) is tested against the fp32 exchange (tests/test_host_cpu.py).
        self.payload_dtype = None if payload_dtype in (None, torch.float32) else payload_dtype
        self.log = []                       # this step's collectives: (kind, bytes, nslices), in issue order
        self.last_buckets = []              # ... of the last completed step (bench.py prints it)

    def _issue(self, t, kind, nslices):
        if self.payload_dtype is not None and t.dtype == torch.float32:
            low = t.to(self.payload_dtype)
            self._staged.append((low, [t]))
            t = low
        self.pending.append(dist.all_reduce(t, async_op=True))
        self.bytes += t.numel() * t.element_size()
        self.log.append((kind, t.numel() * t.element_size(), nslices))

    def _flush_open(self):
        if self._open is not None:
            base, lo, hi, n = self._open
            self._open = None
            self._issue(base.view(-1)[lo:hi], "bucket", n)

    def _flush_small(self):
        if self._small:
            sl, self._small = self._small, []
            pack = torch.cat([t.reshape(-1) for t in sl])
            if self.payload_dtype is not None and pack.dtype == torch.float32:
                pack = pack.to(self.payload_dtype)
            # a LIST: flush() is public, and a slice reduced between a flush() and the wait() (a second backward call, TTA
            # lists, gradient accumulation) starts another pack — every pack in flight is scattered back
            self._staged.append((pack, sl))
            self._issue(pack, "packed", len(sl))

    def _reduce(self, t):
        if not self.active or t.numel() == 0:
            return
        nbytes = t.numel() * t.element_size()
        base = t._base if t._base is not None else t
        span = None
        if t.is_contiguous() and base.is_contiguous() and base.dtype == t.dtype:
            lo = (t.data_ptr() - base.data_ptr()) // t.element_size()
            span = (lo, lo + t.numel())
        o = self._open
        if span is not None and o is not None and o[0] is base and (span[1] == o[1] or span[0] == o[2]):
            o[1], o[2], o[3] = min(o[1], span[0]), max(o[2], span[1]), o[3] + 1       # adjacent: extend the open bucket
        elif nbytes < self.small_bytes or span is None:
            if span is None and nbytes >= self.small_bytes:                           # a large strided tensor: on its own
                self._issue(t, "bucket", 1)
            else:
                self._small.append(t)
            return
        else:
            self._flush_open()
            self._open = [base, span[0], span[1], 1]
        o = self._open
        if (o[2] - o[1]) * t.element_size() >= self.bucket_bytes:
            self._flush_open()

    def vit_hook(self, vit):
        """hook for VisionTransformer.grad_ready_hook: called per block, last block first."""
        def fn(lo, hi):
            self.flush_temporal()           # the temporal backward is complete once the ViT backward starts
            self._reduce(vit.flat.grad[lo:hi])
        fn.active = self.active             # an idle hook (world of one) does not ask for early gradients: vit.py, R6.8
        return fn

    @staticmethod
    def temporal_ranges(model, T):
        """Touched slices of fullModel's flat gradient buffer on the Prototypes path with at most T frames per stream:
        linear (+ importance_function with -il), frame_cls, position rows 0..T-1, the frame-encoder layers.  Everything
        else (clip_*, transEncoderClip, MIL heads, linear2, position rows >= T) never receives a gradient
        (the reference's disabled DDP needed find_unused_parameters=True for them, prepare_model.py:549)."""
        f = model.flat
        first_clip = "transEncoderClip.layers.0.self_attn.in_proj_weight"
        head_end = (f.offsets["linearB.bias"] if "linearB.bias" in f.offsets else f.offsets["linear.bias"]) + 256
        r = [(f.offsets["linear.weight"], head_end),                                  # linear (+ linearB, multi-domain)
             (f.offsets["frame_cls"], f.offsets["frame_cls"] + 384),
             (f.offsets["frame_pos_embeddings.0"], f.offsets["frame_pos_embeddings.0"] + T * 384),
             (f.offsets["transEncoderFrame.layers.0.self_attn.in_proj_weight"], f.offsets[first_clip])]
        if getattr(model, "importance_loss", False):
            r.append((f.offsets["importance_function.weight"], f.offsets["importance_function.bias"] + 4))
        return r

    def temporal_hook(self, model, T=None):
        """hook for fullModel.grad_ready_hook.  It fires once per backward CALL (several per step with TTA list inputs
        or gradient accumulation), so it only marks the model: the exchange itself is issued ONCE per step, when the
        ViT backward starts (vit_hook) or at wait(), over the slices touched by the longest stream seen (RGB and flow
        lengths may differ)."""
        def fn(lo, hi):
            self._temporal = (model, T)
        return fn

    def flush_temporal(self):
        if self._temporal is None:
            return
        model, T = self._temporal
        self._temporal = None
        touched = getattr(model, "_touched_T", 0)
        model._touched_T = 0
        # every rank must hand all_reduce the SAME slices.  A caller that knows the stream length common to ALL ranks
        # passes T (bench.py; trainModel agrees on it with one MAX all-reduce before the first epoch); without it all 2000
        # position rows go (3 MB).  A longer stream than promised on this rank alone would make the collective sizes
        # rank-dependent (a hang or corruption under RCCL), so it is an error, not a silent switch.
        if T is not None and touched > T:
            raise RuntimeError(f"GradSync: a stream of {touched} frames received gradients but the exchange was set up "
                               f"for at most {T} (every rank must all-reduce identical slices)")
        Tmax = T if T is not None else 2000
        for a, b in self.temporal_ranges(model, Tmax):
            self._reduce(model.flat.grad[a:b])

    def reduce_params(self, params):
        for p in params:
            if p.grad is not None:
                self._reduce(p.grad)

    def broadcast_initial_state(self, tensors, src=0):
        """Replicas must START from the same weights: only gradients are exchanged afterwards.  Broadcasts every tensor
        (flat parameter buffers, prototypes) from rank `src`; the caller refreshes its bf16 / transposed shadows."""
        if not self.active:
            return
        for t in tensors:
            dist.broadcast(t, src=src)

    def agree_max(self, value, device):
        """MAX of a host integer over the ranks (stream lengths, stop flags)."""
        if not self.active:
            return int(value)
        t = torch.tensor([int(value)], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return int(t.item())

    def mean_scalar(self, value, device):
        """Mean of a host float over the ranks (the validation loss every rank bases its stop decision on)."""
        if not self.active:
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64, device=device)
        dist.all_reduce(t)
        return float(t.item()) / self.world

    def flush(self):
        """Issue whatever is still open: the last bucket and the packed small slices."""
        self.flush_temporal()
        self._flush_open()
        self._flush_small()

    def wait(self):
        self.flush()
        for w in self.pending:
            w.wait()
        self.pending = []
        staged, self._staged = self._staged, []
        for pack, sl in staged:                          # scatter the staged sums back into their slices (casts up)
            o = 0
            for t in sl:
                t.copy_(pack[o:o + t.numel()].view_as(t))
                o += t.numel()
        self.last_buckets, self.log = self.log, []
        n, self.bytes = self.bytes, 0
        return n


# --------------------------------------------------------------------------------------------------------------
# Inference shards without an exchange step: frames of a video and the windows of a video are independent, so
# every rank takes a contiguous range and the results are gathered in rank order; rank 0 writes the files (train.py:98).
def shard_range(n, rank, world):
    """Contiguous balanced split of range(n): the first n % world ranks get one more."""
    q, r = divmod(int(n), int(world))
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def init_from_env():
    """(rank, world, local_rank) from the torch.distributed.run environment; starts the process group when world > 1
    (backend nccl = RCCL on a GPU box, gloo otherwise).  World 1: nothing is initialised."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        backend = os.environ.get("SAIS_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:                            # gloo: host-side exchange; the ranks may share a GPU (tests on a 1-GPU box)
            if torch.cuda.is_available():
                local %= torch.cuda.device_count()
                torch.cuda.set_device(local)
            dist.init_process_group("gloo", rank=rank, world_size=world)
    elif world > 1 and torch.cuda.is_available():
        local %= torch.cuda.device_count()
    return rank, world, local


def gather_in_rank_order(obj, world):
    """[obj of rank 0, obj of rank 1, ...] on every rank (host objects: CPU tensors, lists).  World 1: [obj]."""
    if world == 1:
        return [obj]
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out
