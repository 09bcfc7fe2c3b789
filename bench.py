#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): frames/s of ViT-S/16 fwd+bwd on synthetic 224x224, 32-frame
clips — one "step" = one full training step of the SAIS hot path on one batch of B=8 clips per GPU:

    256 frames -> ViT-S/16 (12 blocks) -> 4-layer temporal encoder -> prototype (SupCon) loss
    -> backward through head, temporal encoder AND ViT -> (DP: RCCL all-reduce, overlapped) -> SGD step

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  `value` = frames/s over all ranks with inputs resident in HBM;
`roofline` = the dominant MFMA kernel's algorithmic FLOP/s (HIP events on the launch stream, measured in a
separate instrumented pass after the timed region) against the dense bf16 MFMA peak;
`cpu_baseline` = the CPU oracle (oracle/, the validated restatement of the reference) timed on this
host's cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

MFMA_PEAK_TFLOPS = 2500.0        # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md (2:1-sparse figure NOT used)
FLOP_PER_FRAME_FWD_BWD = 27.475e9   # 3 x 9.197 GF - 0.1156 GF (no dX for pixels)
FLOP_TEMPORAL_PER_CLIP = 3 * 0.57764e9


def build(dev, B, T, C, lr):
    import synth
    from sais_amd.optim import SGD
    from sais_amd.temporal import fullModel
    from sais_amd.vit import vit_small
    torch.manual_seed(0)                                  # identical initial weights on every rank
    vit = vit_small(patch_size=16, drop_path_rate=0.0)    # constructor init (trunc-normal .02), seed 0
    vit = vit.to(dev).train()
    model = fullModel('reps', C, 'in_vs_out', 384, 'ViT', modalities='RGB').to(dev).train()
    protos = torch.nn.ParameterDict({str(c): torch.nn.Parameter(torch.rand(1, 256, device=dev)) for c in range(C)})
    opt = SGD(list(vit.parameters()) + list(model.parameters()) + list(protos.values()), lr=lr, engines=[vit, model])
    return vit, model, protos, opt


def make_step(vit, model, protos, opt, sync, frames, pad, labels, B, T, world, dist_on=False):
    from sais_amd.loss import calcNCELoss
    from sais_amd.loss import label_columns
    names = [f"v_{i}" for i in range(B)]
    lens = [T] * B
    labels = label_columns(labels, protos, frames.device)       # static device tensor (graph-capturable)

    def step():
        opt.zero_grad()
        reps = vit(frames).view(B, 1, T, 384)
        emb, attn = model(reps, None, lens, None, 'Prototypes', pad, None, None)
        loss = calcNCELoss(0, emb, labels, names, protos, None)
        loss.backward()
        if dist_on:
            sync.reduce_params(protos.values())
            sync.wait()
        opt.step(grad_scale=1.0 / world)
        return loss
    return step


def cpu_baseline(T, C, threads):
    """CPU oracle (fp32 torch restatement of the reference, pinned to its golden vectors) on a bounded sample:
    one 32-frame clip, fwd + bwd + SGD, same model."""
    import synth
    from oracle import sais_oracle as O
    torch.set_num_threads(threads)
    vsd = {k: v.clone().requires_grad_(True) for k, v in synth.vit_state_dict(seed=0).items()}
    tsd = {k: v.clone().requires_grad_(True) for k, v in synth.temporal_state_dict(seed=1).items()}
    pr = {k: v.clone().requires_grad_(True) for k, v in synth.prototypes(2, C).items()}
    clips = synth.clips(seed=0, B=1, T=T)
    pad = synth.padding_mask([T])
    lab = synth.labels(seed=0, B=1, nclasses=C)

    def step():
        _, emb, _ = O.e2e_forward(vsd, tsd, clips, None, pad, "RGB")
        loss = O.nce_loss(emb, lab, pr)
        loss.backward()
        with torch.no_grad():
            for d in (vsd, tsd, pr):
                for p in d.values():
                    if p.grad is not None:
                        p -= 0.1 * p.grad
                        p.grad = None
    t0 = time.time()
    step()                                                   # warm-up (also sizes the sample)
    warm = time.time() - t0
    budget = 20.0                                            # seconds of timed CPU work
    nmax = max(1, min(10, int(budget / max(warm, 1e-3))))
    n, t0 = 0, time.time()
    while n < nmax:
        step()
        n += 1
    dt = (time.time() - t0) / n
    return dict(value=round(T / dt, 2), unit="frames/s", cores=threads, kind="port",
                sample=f"1 clip x {T} frames (B=1), fwd+bwd+SGD, fp32, {n} timed steps after 1 warm-up, "
                       f"torch {torch.__version__} CPU oracle")


def load_pmc_traffic(kernel):
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(path):
        try:
            return json.load(open(path)).get(kernel)
        except Exception:
            return None
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--clips", type=int, default=8, help="clips per GPU (BASELINE config 2: 8)")
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="issue the ~700 launches of a step eagerly instead of "
                                                              "replaying the captured hipGraph (N=1 only)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # SAIS_BENCH_FORCE_DIST=1 (with torchrun --nproc-per-node 1) takes the distributed code path on a 1-GPU box:
    # RCCL init, gradient all-reduce from the backward hooks, barriers, max-over-ranks timing
    dist_on = world > 1 or (os.environ.get("SAIS_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if dist_on:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import synth
    from sais_amd import ops
    B, T, C = args.clips, args.frames, 2
    vit, model, protos, opt = build(dev, B, T, C, lr=0.1)
    frames = synth.clips(seed=rank, B=B, T=T).view(B * T, 3, 224, 224).to(dev)     # resident in HBM
    pad = synth.padding_mask([T] * B).to(dev)
    labels = synth.labels(seed=rank, B=B, nclasses=C)
    from sais_amd.parallel import GradSync
    sync = GradSync(world, active=dist_on)
    step = make_step(vit, model, protos, opt, sync, frames, pad, labels, B, T, world, dist_on)
    vit(frames[:2])                                          # builds the flat buffers
    model._engine(dev)
    vit.grad_ready_hook = sync.vit_hook(vit)
    model.grad_ready_hook = sync.temporal_hook(model, T)

    eager_step = step
    use_graph = not dist_on and not args.no_graph
    if use_graph:                                            # DP runs eagerly (RCCL collectives from hooks)
        from sais_amd.graph import GraphedStep
        step = GraphedStep(eager_step, warmup=2)
    for _ in range(args.warmup):
        loss = step()

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    if dist_on:
        tmax = torch.tensor([dt], device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()

    # instrumented pass (outside the timed region): HIP events around every MFMA kernel launch
    roof = None
    if rank == 0:
        ops.TIMER = ops.KernelTimer()
        for _ in range(2):
            eager_step()
        torch.cuda.synchronize()
        summ = ops.TIMER.summary()
        ops.TIMER = None
        kern = max(summ, key=lambda k: summ[k]["total_ms"])
        k = summ[kern]
        ach = k["flops"] / (k["avg_ms"] * 1e-3) / 1e12
        pmc = load_pmc_traffic(kern)
        roof = dict(bound="mfma", kernel=kern, achieved=round(ach, 1), peak=MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                    frac=round(ach / MFMA_PEAK_TFLOPS, 4),
                    traffic=(pmc or {}).get("hbm_bytes_per_launch"),      # PMC (profiles/pmc_traffic.json), bytes/launch
                    algorithmic_bytes=int(k["bytes"]),
                    hbm_gbps_algorithmic=round(k["bytes"] / (k["avg_ms"] * 1e-3) / 1e9, 1),
                    avg_launch_us=round(k["avg_ms"] * 1e3, 1), launches_per_step=k["launches"] // 2,
                    all_kernels={n: dict(ms_per_step=round(v["total_ms"] / 2, 3),
                                         tflops=round(v["flops"] / (v["avg_ms"] * 1e-3) / 1e12, 1),
                                         gbps=round(v["bytes"] / (v["avg_ms"] * 1e-3) / 1e9, 1)) for n, v in summ.items()})
    if dist_on:
        dist.barrier()

    if rank == 0:
        fps = world * B * T * args.steps / dt
        step_flops = B * T * FLOP_PER_FRAME_FWD_BWD + B * FLOP_TEMPORAL_PER_CLIP
        out = {
            "metric": "frames/sec ViT-S/16 fwd+bwd, 224x224 32-frame clips",
            "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"BASELINE config 2: ViT-S/16 + 4-layer temporal encoder + SupCon prototype loss, "
                                   f"fwd+bwd+SGD, {B} clips x {T} frames x 224x224 per GPU (global {world * B} clips), "
                                   f"random-init weights, RGB stream", "clips_per_gpu": B, "frames_per_clip": T,
                       "parallelism": f"dp{world}", "launch": "hipGraph replay" if use_graph else "eager"},
            "step_tflops": round(step_flops * world * args.steps / dt / 1e12, 1),
            "frac_of_mfma_roofline": round(step_flops * args.steps / dt / 1e12 / MFMA_PEAK_TFLOPS, 4),
            "loss": round(float(loss.detach()), 6),
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                avail = len(os.sched_getaffinity(0))
            except AttributeError:
                avail = os.cpu_count() or 1
            # fp32 ViT-S on one 32-frame clip stops scaling past a few dozen threads (and oversubscribed
            # hosts get much slower), so use at most 32 of the host's cores; `cores` reports what was used
            out["cpu_baseline"] = cpu_baseline(T, C, max(1, min(avail, 32)))
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
