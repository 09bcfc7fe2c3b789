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
    def __init__(self, world=None, active=None):
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.active = self.world > 1 if active is None else active     # active with world 1: exercises the path
        self.pending = []
        self.bytes = 0                      # bytes handed to all_reduce since the last wait()
        self._temporal = None               # fullModel whose touched slices still have to be exchanged this step

    def _reduce(self, t):
        if self.active and t.numel() > 0:
            self.pending.append(dist.all_reduce(t, async_op=True))
            self.bytes += t.numel() * t.element_size()

    def vit_hook(self, vit):
        """hook for VisionTransformer.grad_ready_hook: called per block, last block first."""
        def fn(lo, hi):
            self.flush_temporal()           # the temporal backward is complete once the ViT backward starts
            self._reduce(vit.flat.grad[lo:hi])
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

    def wait(self):
        self.flush_temporal()
        for w in self.pending:
            w.wait()
        self.pending = []
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
