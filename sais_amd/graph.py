"""hipGraph capture of a fixed-shape training / inference step.

One SAIS training step is ~700 kernel launches issued from Python; captured once into a hipGraph
(torch.cuda.CUDAGraph is the hipGraph wrapper on ROCm) the whole step — zero_grad, ViT forward, temporal
encoder, loss, both backward passes, fused SGD — replays as one launch with no inter-kernel host gaps.
Everything the kernels touch is static: the flat parameter / gradient buffers, the input tensors captured by
the closure, and the activations torch allocates from the graph's private pool."""
import torch


class GraphedStep:
    def __init__(self, fn, warmup=3, capture_error_mode="global"):
        """fn() -> tensor (e.g. the loss); must read its inputs from tensors that outlive the graph.
        capture_error_mode "thread_local": for steps that contain RCCL collectives (ProcessGroupNCCL's watchdog thread
        makes HIP calls of its own while this thread captures)."""
        self.fn = fn
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):                       # allocator, LDS attributes, grad buffers, bf16 shadows
                fn()
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode=capture_error_mode):
            self.out = fn()

    def __call__(self):
        self.graph.replay()
        return self.out
