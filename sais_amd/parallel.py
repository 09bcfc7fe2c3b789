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
import torch.distributed as dist


class GradSync:
    def __init__(self, world=None, active=None):
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.active = self.world > 1 if active is None else active     # active with world 1: exercises the path
        self.pending = []
        self.bytes = 0

    def _reduce(self, t):
        if self.active and t.numel() > 0:
            self.pending.append(dist.all_reduce(t, async_op=True))
            self.bytes += t.numel() * t.element_size()

    def vit_hook(self, vit):
        """hook for VisionTransformer.grad_ready_hook"""
        def fn(lo, hi):
            self._reduce(vit.flat.grad[lo:hi])
        return fn

    @staticmethod
    def temporal_ranges(model, T):
        """Touched slices of fullModel's flat gradient buffer on the Prototypes path with T frames:
        frame_cls, linear.{weight,bias}, position rows 0..T-1, the 4 frame-encoder layers."""
        f = model.flat
        first_clip = "transEncoderClip.layers.0.self_attn.in_proj_weight"
        return [(f.offsets["frame_cls"], f.offsets["frame_cls"] + 384),
                (f.offsets["linear.weight"], f.offsets["linear.bias"] + 256),
                (f.offsets["frame_pos_embeddings.0"], f.offsets["frame_pos_embeddings.0"] + T * 384),
                (f.offsets["transEncoderFrame.layers.0.self_attn.in_proj_weight"], f.offsets[first_clip])]

    def temporal_hook(self, model, T):
        def fn(lo, hi):
            for a, b in self.temporal_ranges(model, T):
                self._reduce(model.flat.grad[a:b])
        return fn

    def reduce_params(self, params):
        for p in params:
            if p.grad is not None:
                self._reduce(p.grad)

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []
