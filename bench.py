#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): frames/s of ViT-S/16 fwd+bwd on synthetic 224x224, 32-frame
clips — one "step" = one full training step of the SAIS hot path on one batch of B=8 clips per GPU:

    256 frames -> ViT-S/16 (12 blocks) -> 4-layer temporal encoder -> prototype (SupCon) loss
    -> backward through head, temporal encoder AND ViT -> (DP: RCCL all-reduce, overlapped) -> SGD step

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no torchrun environment: this process touches no GPU, starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child (one rank per GPU over RCCL)
and exits with its code; fewer than N visible GPUs, or a torchrun WORLD_SIZE that differs from --gpus, is an
error (non-zero exit), never a silent 1-rank run.

Prints ONE JSON line on rank 0.  `value` = frames/s over all ranks with inputs resident in HBM;
`roofline` = the MFMA kernel with the largest total time: algorithmic FLOP/s from HIP events on the launch
stream (raw event intervals, measured in a separate instrumented pass after the timed region) against the
dense bf16 MFMA peak; `cpu_baseline` = the CPU oracle (oracle/, the validated restatement of the reference)
timed on this host's cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

MFMA_PEAK_TFLOPS = 2500.0        # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md (2:1-sparse figure NOT used)
HBM_PEAK_GBPS = 8000.0           # HBM3E spec (same guide; ~6.3 TB/s achievable)
FLOP_PER_FRAME_FWD_BWD = 27.475e9   # 3 x 9.197 GF - 0.1156 GF (no dX for pixels)
# The last ViT block runs on the CLS rows / the CLS query only (sais_amd.vit, prune_last_block: forward() returns x[:, 0], so
# the other rows of that block feed nothing): its proj, MLP and query-side attention products are not executed.  Per frame:
# 3 x (proj .058098 + fc1 .232391 + fc2 .232391) + attention (fwd .059610, bwd 2.5 x) + the query third of the qkv GEMM
# and of its weight gradient (2 x .058098; its dX still runs, on zeros) = 1.893 GF.  The roofline fractions below are
# computed from the EXECUTED flops, so the pruning does not inflate them.
FLOP_PRUNED_PER_FRAME = 3 * (0.058098e9 + 2 * 0.232391e9) + 3.5 * 0.059610e9
FLOP_PRUNED_Q_PER_FRAME = 2 * 0.058098e9       # SAIS_VIT_PRUNE_Q (default on)
FLOP_TEMPORAL_PER_CLIP = 3 * 0.57764e9
HBM_BYTES_PER_FRAME = 133e6      # SURVEY §8d / BASELINE.md §4: fused bf16 plan, fwd+bwd


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)          # 50 x ~13 ms: a timed region of ~0.65 s (round 3's default: 0.27 s)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--clips", type=int, default=8, help="clips per GPU (BASELINE config 2: 8)")
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="issue the launches of a step eagerly instead of "
                                                              "replaying the captured hipGraph")
    ap.add_argument("--two-stream", action="store_true",
                    help="BASELINE config 4: RGB + optical-flow frames, 2 x clips x frames through the ViT, the temporal "
                         "encoder once per stream, streams fused by add (modalities='RGB-Flow', prepare_model.py:412)")
    ap.add_argument("--vit-drop-path", type=float, default=0.1,
                    help="stochastic-depth rate of the ViT in train mode (the reference constructs it with 0.1: "
                         "extract_representations.py:201, main_dino.py:57)")
    ap.add_argument("--workload", choices=("train", "dino", "extract"), default="train",
                    help="train = the headline step above (BASELINE.json); dino = one DINO pre-training step "
                         "(SURVEY §8f-4: main_dino.py defaults, 64 images per GPU, 2 x 224 + 8 x 96 crops, out_dim 65536); "
                         "extract = BASELINE config 5, long-video inference: one 512-frame video per step through the frozen "
                         "ViT (hipGraph, batches of 64), its 34 flow maps, then the 34 sliding windows x 3 TTA versions "
                         "through the temporal encoder with the attention maps exported to the host")
    ap.add_argument("--video-frames", type=int, default=512, help="--workload extract: frames of the synthetic video")
    ap.add_argument("--extract-batch", type=int, default=256,
                    help="--workload extract: frames per hipGraph replay (main.sh asks for 1024, the CLI caps it at 256 = "
                         "50 432 token rows, where the GEMM kernels are at their best)")
    ap.add_argument("--extract-tail", type=int, default=-1,
                    help="--workload extract: frames of the second captured shape that takes the remainder of a video and "
                         "short inputs (the 34 flow maps); 0 = pad the remainder to --extract-batch; -1 (default) = one captured "
                         "shape per distinct remainder, rounded up to an even number of frames")
    ap.add_argument("--window-batch", type=int, default=2, help="--workload extract: windows per batch (main.sh: -bs 2)")
    ap.add_argument("--grad-payload", choices=("fp32", "bf16"), default="fp32",
                    help="N > 1: what the gradient all-reduce carries (bf16 = half the xGMI bytes, sums rounded to bf16)")
    ap.add_argument("--dino-batch", type=int, default=64, help="--workload dino: images per GPU (batch_size_per_gpu)")
    ap.add_argument("--dino-local-crops", type=int, default=8)
    ap.add_argument("--dino-out-dim", type=int, default=65536)
    ap.add_argument("--dino-graph", action="store_true",
                    help="--workload dino: replay forward + loss + backward as one hipGraph (sais_amd.dino.GraphedTrainStep). "
                         "Off by default: the eager DINO step is device-bound and measured 0.5 %% FASTER (LABNOTES R4.3)")
    ap.add_argument("--parity-clips", type=int, default=1,
                    help="clips of the timed batch that also go through the CPU oracle (with the draws the GPU forward "
                         "used) for the in-line parity gate; 0 = skip")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the short BASELINE config 4 (two-stream) and config 5 (long-video extraction) legs that the "
                         "default one-GPU run reports under `variants`")
    ap.add_argument("--sustain-seconds", type=float, default=10.0,
                    help="after the timed region, keep replaying the step for this long and report the settled rate "
                         "(clock under sustained MFMA load); 0 = skip")
    return ap.parse_args()


def spawn_ranks(args):
    """--gpus N > 1 outside torchrun: launch N ranks as CHILD processes (this process never initialises the GPU)."""
    import torch
    have = torch.cuda.device_count()          # counting devices does not initialise HIP on this image
    if have < args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} requested but only {have} GPU(s) are visible; "
                         "refusing to run fewer ranks than asked\n")
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def build(dev, B, T, C, lr, two_stream=False, drop_path=0.1):
    import torch
    from sais_amd.optim import SGD
    from sais_amd.temporal import fullModel
    from sais_amd.vit import vit_small
    torch.manual_seed(0)                                  # identical initial weights on every rank
    vit = vit_small(patch_size=16, drop_path_rate=drop_path)   # constructor init (trunc-normal .02), seed 0
    vit = vit.to(dev).train()
    vit.drop_path_seed = 7919 + int(os.environ.get("RANK", "0"))
    model = fullModel('reps', C, 'in_vs_out', 384, 'ViT', modalities='RGB-Flow' if two_stream else 'RGB').to(dev).train()
    # train mode = the reference's: dropout 0.1 in the temporal encoder (prepare_model.py:75, train.py:59); every rank its
    # own mask stream
    model.dropout_seed = int(os.environ.get("RANK", "0"))
    protos = torch.nn.ParameterDict({str(c): torch.nn.Parameter(torch.rand(1, 256, device=dev)) for c in range(C)})
    opt = SGD(list(vit.parameters()) + list(model.parameters()) + list(protos.values()), lr=lr, engines=[vit, model])
    return vit, model, protos, opt


def make_step(vit, model, protos, opt, sync, frames, pad, labels, B, T, world, dist_on=False, comm_events=None,
              two_stream=False):
    import torch
    from sais_amd.loss import calcNCELoss
    from sais_amd.loss import label_columns
    names = [f"v_{i}" for i in range(B)]
    lens = [T] * B
    labels = label_columns(labels, protos, frames.device)       # static device tensor (graph-capturable)

    def step():
        opt.zero_grad()
        if two_stream:       # frames = [RGB clips | flow clips]: one ViT pass over 2 B T frames, one encoder pass per stream
            reps = vit(frames).view(2, B, 1, T, 384)
            emb, attn = model(reps[0], reps[1], lens, lens, 'Prototypes', pad, pad, None)
        else:
            reps = vit(frames).view(B, 1, T, 384)
            emb, attn = model(reps, None, lens, None, 'Prototypes', pad, None, None)
        loss = calcNCELoss(0, emb, labels, names, protos, None)
        loss.backward()
        if dist_on:
            sync.reduce_params(protos.values())
            sync.flush()                                  # the last open bucket + the packed small slices go out now
            if comm_events is not None and comm_events.enabled:   # compute-stream stall for the collectives = exposed comm
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                nlaunch = len(sync.pending)
                nbytes = sync.wait()
                e1.record()
                comm_events.recs.append((e0, e1, nbytes, nlaunch))
            else:
                sync.wait()
        opt.step(grad_scale=1.0 / world)
        return loss
    return step


class CommEvents:
    """HIP-event brackets around the gradient-exchange join; only armed in the eager instrumented passes."""

    def __init__(self):
        self.enabled, self.recs = False, []


# ------------------------------------------------------------------------------------------ CPU baseline
def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(T, C, threads, weights, Bfull=8):
    """CPU oracle (fp32 torch restatement of the reference, pinned to its golden vectors) on bounded samples of the
    configurations BASELINE.md §3 lists, with the TIMED model's own weights (`weights` = its ViT / temporal state dicts and
    prototypes on the host).  `value` = config 2's model on one 32-frame clip, fwd+bwd+SGD (the same figure as round 1);
    `variants` holds config 1 and the B=8 / forward-only legs."""
    import torch
    import synth
    from oracle import sais_oracle as O
    torch.set_num_threads(threads)
    w_vit, w_tmp, w_pro = weights

    def leg(B, Tn, nlayers, train, budget_s, max_steps):
        vsd = {k: v.clone().requires_grad_(train) for k, v in w_vit.items()}
        tsd = {k: v.clone().requires_grad_(train) for k, v in w_tmp.items() if "frame_pos_embeddings" not in k or
               int(k.rsplit(".", 1)[1]) < Tn}                 # the oracle reads position rows 0..T-1 only
        pr = {k: v.clone().requires_grad_(train) for k, v in w_pro.items()}
        clips = synth.clips(seed=0, B=B, T=Tn)
        pad = synth.padding_mask([Tn] * B)
        lab = synth.labels(seed=0, B=B, nclasses=C)

        def step():
            if not train:
                with torch.no_grad():
                    _, emb, _ = O.e2e_forward(vsd, tsd, clips, None, pad, "RGB", nlayers=nlayers)
                    O.cosine_logits(emb, pr)
                return
            _, emb, _ = O.e2e_forward(vsd, tsd, clips, None, pad, "RGB", nlayers=nlayers)
            loss = O.nce_loss(emb, lab, pr)
            loss.backward()
            with torch.no_grad():
                for d in (vsd, tsd, pr):
                    for p in d.values():
                        if p.grad is not None:
                            p -= 0.1 * p.grad
                            p.grad = None
        t0 = time.time()
        step()                                               # warm-up (also sizes the sample)
        warm = time.time() - t0
        n_max = max(1, min(max_steps, int(budget_s / max(warm, 1e-3))))
        n, t0 = 0, time.time()
        while n < n_max:
            step()
            n += 1
        dt = (time.time() - t0) / n
        return round(B * Tn / dt, 2), n

    # the headline leg IS the benchmarked configuration (B clips of T frames, fwd+bwd+SGD): 3 timed steps after one warm-up
    # (~15 s each on 32 cores: BASELINE.md §3 asks for >= 3); the other legs are a few seconds each
    main, n_main = leg(Bfull, T, 4, True, 1e9, 3)
    variants = {f"config2_B{Bfull}_T{T}_fwd_bwd_sgd": {"frames_per_s": main, "timed_steps": n_main}}     # = `value`
    for name, (B, Tn, nl, train, budget, mx) in {
            "config1_B1_T16_1layer_fwd": (1, 16, 1, False, 1.5, 3),
            "config1_B1_T16_1layer_fwd_bwd": (1, 16, 1, True, 2.5, 3),
            f"config2_B1_T{T}_fwd": (1, T, 4, False, 2.0, 3),
            f"config2_B1_T{T}_fwd_bwd_sgd": (1, T, 4, True, 6.0, 4)}.items():
        v, n = leg(B, Tn, nl, train, budget, mx)
        variants[name] = {"frames_per_s": v, "timed_steps": n}
    return dict(value=main, unit="frames/s", cores=threads, kind="port", cpu=cpu_model_name(),
                sample=f"{Bfull} clips x {T} frames (the timed batch shape), 4-layer temporal encoder, fwd+bwd+SGD, fp32, "
                       f"{n_main} timed steps after 1 warm-up, torch {torch.__version__} CPU oracle",
                variants=variants)


def dino_cpu_baseline(args, threads, sample=2):
    """The oracle's DINO step (oracle/dino_oracle.py, fp32 torch on the host cores) on `sample` images with the same
    crop counts and out_dim; value = sample / time of one step (the second of two)."""
    import numpy as np
    import torch
    import synth
    from oracle import dino_oracle as do
    torch.set_num_threads(threads)
    sd = {"backbone." + k: v for k, v in synth.vit_state_dict(seed=20).items()}
    sd.update({"head." + k: v for k, v in synth.dino_head_state_dict(seed=21, out_dim=args.dino_out_dim).items()})
    st = do.TrainState(sd, dtype=torch.float32)
    crops = synth.dino_crops(seed=1, B=sample, n_local=args.dino_local_crops)
    sched = np.full(4, 1e-4)
    dt = 0.0
    for it in range(2):
        t0 = time.perf_counter()
        do.train_step(st, crops, it, 1, sched, sched, np.full(4, 0.996), np.full(4, 0.04), 3.0, 1, args.dino_local_crops)
        dt = time.perf_counter() - t0
    return dict(value=round(sample / dt, 3), unit="images/s", cores=threads, kind="port", cpu=cpu_model_name(),
                sample=f"one step on {sample} images (2 x 224 + {args.dino_local_crops} x 96 crops each), out_dim "
                       f"{args.dino_out_dim}, fp32, second of two steps, torch {torch.__version__} CPU oracle")


def dino_main(args):
    """--workload dino: K steps of sais_amd.dino.train_step (teacher forward on the 2 global crops, student forward +
    backward on all crops, DINOLoss + centre, per-parameter clipping, AdamW, EMA teacher) on synthetic crops resident in
    HBM; one rank per GPU, gradients and the [1, out_dim] centre all-reduced over RCCL when N > 1 (weak scaling)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # SAIS_BENCH_FORCE_DIST=1 under torchrun --nproc-per-node 1: RCCL with a world of one, every all-reduce of the DP path issued
    force = os.environ.get("SAIS_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ
    dist_on = world > 1 or force
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from sais_amd import dino, ops
    dino.FORCE_SYNC = force
    B, nl, out_dim = args.dino_batch, args.dino_local_crops, args.dino_out_dim
    torch.manual_seed(0)                                             # same initial weights on every rank
    student, teacher = dino.build_student_teacher(out_dim=out_dim, drop_path_rate=args.vit_drop_path, device=dev)
    loss_mod = dino.DINOLoss(out_dim, nl + 2, 0.04, 0.04, 0, 100).to(dev)
    opt = dino.DINOOptimizer(student, teacher)
    n = args.warmup + args.steps + 2
    lr_s = dino.cosine_scheduler(0.0005 * B * world / 256.0, 1e-6, 100, n)
    wd_s = dino.cosine_scheduler(0.04, 0.4, 100, n)
    mom_s = dino.cosine_scheduler(0.996, 1, 100, n)
    g = torch.Generator(device=dev).manual_seed(1 + rank)
    images = [torch.randn(B, 3, 224, 224, device=dev, generator=g) for _ in range(2)] + \
             [torch.randn(B, 3, 96, 96, device=dev, generator=g) for _ in range(nl)]

    def eager_step(it):
        return dino.train_step(student, teacher, loss_mod, opt, images, it, 1, lr_s, wd_s, mom_s, clip_grad=3.0,
                               freeze_last_layer=1)[0]

    # --dino-graph: everything up to the optimizer tail replayed as one hipGraph (sais_amd.dino.GraphedTrainStep)
    step, launch = eager_step, "eager"
    if args.dino_graph:
        try:
            graphed = dino.GraphedTrainStep(student, teacher, loss_mod, opt, images, clip_grad=3.0)
            graphed(0, 1, lr_s, wd_s, mom_s, 1)                      # captures (and runs iteration 0)
            step = lambda it: graphed(it, 1, lr_s, wd_s, mom_s, 1)[0]
            launch = "hipGraph replay (forward, loss, backward, gradient exchange, norms) + eager optimizer tail"
        except Exception as e:                                      # e.g. a RCCL build that refuses stream capture
            sys.stderr.write(f"bench.py[rank {rank}]: graph capture of the DINO step failed ({type(e).__name__}: {e}); "
                             "running eagerly\n")
            torch.cuda.synchronize()

    for it in range(args.warmup):
        loss = step(it)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(args.steps):
        loss = step(args.warmup + it)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], device=dev)
    if dist_on:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    ms = dt.item() / args.steps * 1e3
    lv = loss.item()
    comm_bytes = 4 * (student.backbone.flat.numel + student.head.flat.numel) if dist_on else 0
    ops.TIMER = ops.KernelTimer()                                    # instrumented pass after the timed region (eager)
    eager_step(args.warmup + args.steps)
    torch.cuda.synchronize()
    summ, ops.TIMER = ops.TIMER.summary(), None
    top = max(summ.items(), key=lambda kv: kv[1]["total_ms"])
    tf = top[1]["flops"] / (top[1]["avg_ms"] * 1e-3) / 1e12
    if rank == 0:
        crops = f"2 x 224 + {nl} x 96"
        out = {"metric": "dino_pretrain_images_per_s", "value": round(B * world / (ms * 1e-3), 1), "unit": "images/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": f"DINO ViT-S/16 pre-training step (main_dino.py defaults): {B} images per GPU, "
                                      f"{crops} crops, out_dim {out_dim}, drop_path {args.vit_drop_path}, clip 3.0, AdamW, "
                                      f"EMA teacher", "parallelism": f"dp{world}", "launch": launch,
                          "student_token_rows_per_gpu": B * (2 * 197 + nl * 37), "teacher_token_rows_per_gpu": B * 2 * 197},
               "loss": round(lv, 4), "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
               "roofline": {"kernel": top[0], "bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_PEAK_TFLOPS,
                            "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TFLOPS, 4), "traffic": None,
                            "launches_per_step": top[1]["launches"], "avg_us": round(top[1]["avg_ms"] * 1e3, 1)},
               "kernels_ms_per_step": {k: round(v["total_ms"], 3) for k, v in
                                       sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])[:12]}}
        if dist_on:
            sync = getattr(opt, "_sync", None)
            out["comm"] = {"backend": "nccl (RCCL)", "world": world, "allreduce_bytes_per_step": comm_bytes,
                           "overlapped_with_backward": True, "centre_allreduce_bytes": 4 * out_dim,
                           "grad_sync": type(sync).__name__ if sync is not None else None}
        if world == 1 and not dist_on and not args.no_cpu_baseline:
            try:
                avail = len(os.sched_getaffinity(0))
            except AttributeError:
                avail = os.cpu_count() or 1
            out["cpu_baseline"] = dino_cpu_baseline(args, max(1, min(avail, 32)))    # see the note in main()
        print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


FLOP_PER_FRAME_FWD = 9.197e9          # ViT-S/16 forward, 197 tokens


def extract_cpu_baseline(threads, vsd, tsd, window_batch):
    """Forward-only CPU leg of config 5 on a bounded sample: a 60-frame video (4 flow maps, 4 windows x 3 TTA versions)
    through the oracle's ViT and temporal encoder with the timed model's weights; value = RGB frames per second."""
    import torch
    import synth
    from oracle import sais_oracle as O
    from sais_amd.inference import gesture_windows, sample_window
    torch.set_num_threads(threads)
    N = 60
    frames = synth.clips(seed=5, B=1, T=N)[0]
    flow_frames = synth.clips(seed=6, B=1, T=N // 15)[0]

    def video():
        with torch.no_grad():
            reps = torch.cat([O.vit_forward(vsd, frames[i:i + 30]) for i in range(0, N, 30)])
            freps = O.vit_forward(vsd, flow_frames)
            for s, e in gesture_windows(N):
                xs, fs = sample_window(reps, freps, s, e)
                for x, f in zip(xs, fs):
                    x, f = x.unsqueeze(0), f.unsqueeze(0)
                    O.temporal_forward(tsd, x, f, torch.zeros(1, 1, x.shape[2] + 1, dtype=torch.bool),
                                       torch.zeros(1, 1, f.shape[2] + 1, dtype=torch.bool), "RGB-Flow")
    video()
    n, t0 = 0, time.time()
    while n < 3 or (time.time() - t0 < 8.0 and n < 10):
        video()
        n += 1
    dt = (time.time() - t0) / n
    return dict(value=round(N / dt, 2), unit="frames/s", cores=threads, kind="port", cpu=cpu_model_name(),
                sample=f"a {N}-frame video (+ {N // 15} flow maps, {len(gesture_windows(N))} windows x 3 TTA versions), "
                       f"forward only, fp32, {n} timed passes after 1 warm-up, torch {torch.__version__} CPU oracle")


def extract_main(args):
    """--workload extract = BASELINE config 5 (long-video inference, TP = 1).  One step = one synthetic video resident in
    HBM: N frames (default 512) and their N // 15 flow maps through the frozen ViT-S/16 (extract_representations.py:351-378;
    fixed batches of 64 replayed from one hipGraph), then every sliding window of prepare_dataset.py:1711-1725 (34 of
    them) x 3 test-time-augmentation index sets through the two-stream temporal encoder, embeddings and the [34, 16, 16]
    attention maps copied to the host as the reference saves them (train.py:113-119).  value = RGB frames per second.
    N > 1: every rank runs its own video (independent shards, no collective on the data path: weak scaling)."""
    import torch
    import torch.distributed as dist
    import synth
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    backend = os.environ.get("SAIS_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist_on = world > 1
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    from sais_amd import ops
    from sais_amd.inference import FeatureExtractor, gesture_windows, run_windows
    from sais_amd.temporal import fullModel
    from sais_amd.vit import vit_small
    N, bs = args.video_frames, args.extract_batch
    torch.manual_seed(0)
    vit = vit_small(patch_size=16).to(dev).eval()
    model = fullModel('reps', 2, 'in_vs_out', 384, 'ViT', modalities='RGB-Flow').to(dev).eval()
    frames = synth.clips(seed=5 + rank, B=1, T=N)[0].to(dev)                     # resident in HBM
    flow_frames = synth.clips(seed=1000 + rank, B=1, T=max(1, N // 15))[0].to(dev)
    tail = "fit" if args.extract_tail < 0 else (args.extract_tail or None)
    fx = FeatureExtractor(vit, batch_size=bs, use_graph=not args.no_graph, tail_batch=tail)
    nwin = len(gesture_windows(N))

    def step():
        reps = fx(frames)
        freps = fx(flow_frames)
        out, attention, _ = run_windows(model, reps, freps, videoname="synthetic", batch_size=args.window_batch)
        return reps, freps, out, attention

    for _ in range(max(1, args.warmup)):
        reps, freps, out, attention = step()

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        reps, freps, out, attention = step()
    fence()
    dt = time.perf_counter() - t0
    if dist_on:
        tmax = torch.tensor([dt], device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    if rank != 0:
        if dist_on:
            dist.barrier()
            dist.destroy_process_group()
        return
    # where the step's time goes: the two halves timed separately (same inputs, after the timed region)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(5):
        fx(frames); fx(flow_frames)
    torch.cuda.synchronize()
    ms_vit = (time.perf_counter() - t1) / 5 * 1e3
    t1 = time.perf_counter()
    for _ in range(5):
        run_windows(model, reps, freps, videoname="synthetic", batch_size=args.window_batch)
    torch.cuda.synchronize()
    ms_win = (time.perf_counter() - t1) / 5 * 1e3
    # parity gate: four frames of the timed video and three windows against the CPU oracle, same weights
    from oracle import sais_oracle as O
    from sais_amd.inference import sample_window
    vsd = {k: v.detach().float().cpu() for k, v in vit.state_dict().items()}
    tsd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    idx = sorted({0, min(bs - 1, N - 1), min(bs, N - 1), N - 1})
    with torch.no_grad():
        ref = O.vit_forward(vsd, frames[idx].cpu())
    feat_rel = float((reps[idx].cpu() - ref).abs().max() / ref.abs().max())
    wins = gesture_windows(N)
    rc, fc = reps.cpu(), freps.cpu()
    attn = torch.cat(attention)
    pr = synth.prototypes(2, 2)
    dl = da = 0.0
    for w in sorted({0, 1, nwin - 1}):
        xs, fs = sample_window(rc, fc, *wins[w])
        for v in range(3):
            x, f = xs[v].unsqueeze(0), fs[v].unsqueeze(0)
            with torch.no_grad():
                e_ref, a_ref = O.temporal_forward(tsd, x, f, torch.zeros(1, 1, x.shape[2] + 1, dtype=torch.bool),
                                                  torch.zeros(1, 1, f.shape[2] + 1, dtype=torch.bool), "RGB-Flow")
            dl = max(dl, float((O.cosine_logits(out["reps"][v][w].unsqueeze(0), pr) - O.cosine_logits(e_ref, pr)).abs().max()))
            if v == 0:
                da = max(da, float((attn[w] - a_ref[0]).abs().max()))
    parity = dict(max_abs_feature_rel=round(feat_rel, 6), max_abs_logit=round(dl, 7), max_abs_attn=round(da, 7),
                  tolerance_logit=1e-3, frames_checked=len(idx), windows_checked=len({0, 1, nwin - 1}),
                  what="features of the timed video (hipGraph replay) and the temporal half at the GPU's own features vs "
                       "the fp32 CPU oracle, same weights; logits = cosines against seeded prototypes")
    parity["pass"] = bool(dl <= 1e-3 and da <= 2e-3 and feat_rel <= 2e-2)
    # instrumented pass: HIP events around every MFMA kernel of one eager video (the replayed graph holds the same launches)
    fe = FeatureExtractor(vit, batch_size=bs, use_graph=False, tail_batch=tail)
    fe(frames); fe(flow_frames)                              # untimed eager pass first: a first eager launch of a shape can carry
    torch.cuda.synchronize()                                 # one-off host work (seen once as a 3-ms "attention" launch)
    ops.TIMER = ops.KernelTimer()
    fe(frames); fe(flow_frames)
    torch.cuda.synchronize()
    summ, ops.TIMER = ops.TIMER.summary(), None
    kern = max(summ, key=lambda t: summ[t]["total_ms"])
    k = summ[kern]
    ach = k["flops"] / (k["avg_ms"] * 1e-3) / 1e12
    fps = world * N * args.steps / dt
    nvit = N + max(1, N // 15)
    out_line = {
        "metric": "frames/sec ViT-S/16 long-video inference (extraction + sliding windows + attention export)",
        "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"BASELINE config 5: {N}-frame synthetic video per GPU, frozen ViT-S/16 forward in hipGraph-replayed "
                               f"batches of {bs} (remainders and the {nvit - N} flow maps in "
                               f"{'a batch of their own size' if tail == 'fit' else 'batches of %d' % (tail or bs)}), "
                               f"{nwin} sliding windows x 3 TTA versions through the "
                               f"two-stream 4-layer temporal encoder (window batch {args.window_batch}, main.sh's -bs), embeddings "
                               f"+ attention maps [{nwin},16,16] exported to the host; random-init weights",
                   "video_frames": N, "vit_frames_per_step": nvit, "windows": nwin, "parallelism": f"shards{world}",
                   "launch": "hipGraph replay (ViT) + eager windows" if not args.no_graph else "eager"},
        "split_ms": {"vit_extraction": round(ms_vit, 3), "windows_and_export": round(ms_win, 3)},
        "vit_forward_tflops": round(nvit * FLOP_PER_FRAME_FWD / (ms_vit * 1e-3) / 1e12, 1),
        "frac_of_mfma_roofline": round(nvit * FLOP_PER_FRAME_FWD * args.steps / dt / 1e12 / MFMA_PEAK_TFLOPS, 4),
        "parity": parity,
        "roofline": {"bound": "mfma", "kernel": kern, "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": None, "algorithmic_bytes": int(k["bytes"]),
                     "hbm_gbps_algorithmic": round(k["bytes"] / (k["avg_ms"] * 1e-3) / 1e9, 1),
                     "avg_launch_us": round(k["avg_ms"] * 1e3, 1), "launches_per_step": k["launches"],
                     "timing": "raw HIP-event interval per launch in one eager pass over the video after the timed region",
                     "all_kernels": {n: dict(ms_per_step=round(v["total_ms"], 3), launches_per_step=v["launches"],
                                             avg_us=round(v["avg_ms"] * 1e3, 1),
                                             tflops=round(v["flops"] / (v["avg_ms"] * 1e-3) / 1e12, 1),
                                             gbps=round(v["bytes"] / (v["avg_ms"] * 1e-3) / 1e9, 1)) for n, v in summ.items()}},
    }
    if world == 1 and not args.no_cpu_baseline:
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        out_line["cpu_baseline"] = extract_cpu_baseline(max(1, min(avail, 32)), vsd, tsd, args.window_batch)
    if not parity["pass"]:
        out_line["invalid"] = True
        sys.stderr.write(f"bench.py: PARITY GATE FAILED: {parity}\n")
    print(json.dumps(out_line), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if not parity["pass"]:
        sys.exit(3)


def parity_gate(vit, model, protos, frames, pad, labels, B, T, C, two, nclips):
    """BASELINE.md §3 / perform_training.py:119-127: the TIMED model on the TIMED inputs against the CPU oracle, in the same
    run.  One train-mode forward of the whole batch on the GPU (temporal dropout and ViT DropPath on, as timed); the draws
    it used are exported (per-sample DropPath factors, dropout keep masks regenerated from the forward's RNG state) and
    the first `nclips` clips go through the oracle with the same weights and the same draws.  Returns max-abs deviations
    of the cosine class logits, the returned attention map, the embeddings, and of the per-clip NCE loss."""
    import torch
    from oracle import sais_oracle as O
    from sais_amd.loss import calcNCELoss, cosine_logits_and_probs
    n, S = min(nclips, B), T + 1
    lens = [T] * B
    with torch.no_grad():
        reps = vit(frames)
        if two:
            r = reps.view(2, B, 1, T, 384)
            emb, attn = model(r[0], r[1], lens, lens, 'Prototypes', pad, pad, None)
        else:
            emb, attn = model(reps.view(B, 1, T, 384), None, lens, None, 'Prototypes', pad, None, None)
        sim, _ = cosine_logits_and_probs(emb, protos)
        loss_gpu = float(calcNCELoss(0, emb[:n].contiguous(), labels[:n], [f"v_{i}" for i in range(n)], protos, None))
    fac = None
    if vit.training and vit.drop_path_rate > 0:
        sc = vit.last_droppath_scales.view(2 * vit.depth, -1, 197)[:, :, 0].cpu()          # [24, frames] per-sample factors
        fac = sc
    drop = None
    if model.training and model.dropout_p > 0:
        st = model.last_dropout_state
        drop = {"rgb": [{k: v[:n].cpu() for k, v in lm.items()} for lm in model.dropout_masks(st, B, S, stream=0)]}
        if two:
            drop["flow"] = [{k: v[:n].cpu() for k, v in lm.items()} for lm in model.dropout_masks(st, B, S, stream=1)]
    vsd = {k: v.detach().float().cpu() for k, v in vit.state_dict().items()}
    tsd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    pr = {k: v.detach().float().cpu() for k, v in protos.items()}
    fr = frames.cpu()
    t0 = time.time()
    with torch.no_grad():
        def feats(lo):                                        # oracle ViT on frames [lo, lo + n T) with their DropPath factors
            return O.vit_forward(vsd, fr[lo:lo + n * T], droppath=None if fac is None else fac[:, lo:lo + n * T])
        x = feats(0).view(n, 1, T, 384)
        f = feats(B * T).view(n, 1, T, 384) if two else None
        pc = pad[:n].cpu()
        e_ref, a_ref = O.temporal_forward(tsd, x, f, pc, pc if two else None, "RGB-Flow" if two else "RGB", drop=drop,
                                          p=float(model.dropout_p))
        sim_ref = O.cosine_logits(e_ref, pr)
        loss_ref = float(O.nce_loss(e_ref, labels[:n], pr))
    dl = float((sim[:n].cpu() - sim_ref).abs().max())
    out = dict(max_abs_logit=dl, max_abs_attn=float((attn[:n].cpu() - a_ref).abs().max()),
               max_abs_emb=float((emb[:n].cpu() - e_ref).abs().max()), loss_abs=abs(loss_gpu - loss_ref),
               max_abs_feature_rel=float((reps[:n * T].cpu() - x.view(n * T, 384)).abs().max() / x.abs().max()),
               tolerance_logit=1e-3, clips_checked=n, frames_checked=n * T * (2 if two else 1),
               oracle_seconds=round(time.time() - t0, 1),
               what="timed model (same weights) on the timed inputs, train mode, temporal dropout "
                    f"{float(model.dropout_p) if model.training else 0.0} + ViT DropPath "
                    f"{vit.drop_path_rate if vit.training else 0.0} with the draws of the GPU forward fed to the fp32 CPU "
                    "oracle (oracle/sais_oracle.py, pinned to the reference's golden vectors)")
    out = {k: (round(v, 7) if isinstance(v, float) else v) for k, v in out.items()}
    out["pass"] = bool(dl <= 1e-3)
    return out


def kernel_source_hash(root=ROOT):
    """sha256 over the kernel sources (sais_amd/csrc/*.hip, *.hpp, Makefile, include/*.h), in name order.  Counter files under
    profiles/ carry the hash of the tree they were collected on (tools/pmc_report.py), so a later kernel edit cannot leave a stale
    `traffic` in the benchmark line: load_pmc() returns the file only while the hashes agree."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "sais_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "sais_amd", "csrc", "*.hpp"))
                   + glob.glob(os.path.join(root, "include", "*.h")) + [os.path.join(root, "sais_amd", "csrc", "Makefile")])
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load_pmc(name, root=ROOT):
    """(counters, stale): the parsed profiles/<name>, and whether it describes OTHER kernel sources than this tree's (no hash in
    its _provenance counts as stale)."""
    path = os.path.join(root, "profiles", name)
    if os.path.exists(path):
        try:
            d = json.load(open(path))
        except Exception:
            return None, False
        return d, (d.get("_provenance") or {}).get("kernel_source_hash") != kernel_source_hash(root)
    return None, False


def run_variants(args):
    """BASELINE configs 4 and 5 beside the headline: two short child runs of this file (`--two-stream`, `--workload extract`; 5
    timed steps each, their own parity gates on), condensed.  Children, not in-process: each leg builds its own models and graphs
    and leaves nothing behind in this process; they start after the headline's timed and sustained regions."""
    base = [sys.executable, os.path.abspath(__file__), "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-variants"]
    legs = {"config4_two_stream": base + ["--two-stream", "--sustain-seconds", "0"],
            "config5_long_video": base + ["--workload", "extract"]}
    out = {}
    for name, cmd in legs.items():
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            line = json.loads(r.stdout.strip().splitlines()[-1])
            par = line.get("parity") or {}
            v = {"value": line["value"], "unit": line["unit"], "ms_per_step": line["ms_per_step"], "steps": line["steps"],
                 "workload": line["config"]["workload"], "frac_of_mfma_roofline": line.get("frac_of_mfma_roofline"),
                 "roofline": {k: line["roofline"].get(k) for k in ("kernel", "achieved", "frac", "avg_launch_us")},
                 "parity": {k: par.get(k) for k in ("max_abs_logit", "max_abs_attn", "max_abs_feature_rel", "tolerance_logit", "pass")},
                 "exit_code": r.returncode, "wall_s": round(time.perf_counter() - t0, 1)}
            if "split_ms" in line:
                v["split_ms"] = line["split_ms"]
            if line.get("invalid"):
                v["invalid"] = True
        except Exception as e:                                   # a failed leg is reported, it does not void the headline
            v = {"error": repr(e)[:300], "wall_s": round(time.perf_counter() - t0, 1)}
        out[name] = v
    return out


# ------------------------------------------------------------------------------------------ main
def main():
    args = parse_args()
    force_dist = os.environ.get("SAIS_BENCH_FORCE_DIST") == "1"
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(spawn_ranks(args))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus and not force_dist:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")
    if args.workload == "dino":
        return dino_main(args)
    if args.workload == "extract":
        return extract_main(args)

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # SAIS_BENCH_FORCE_DIST=1 (with torchrun --nproc-per-node 1) takes the distributed code path on a 1-GPU box:
    # RCCL init, gradient all-reduce from the backward hooks, barriers, max-over-ranks timing
    dist_on = world > 1 or (force_dist and "RANK" in os.environ)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # SAIS_BENCH_BACKEND=gloo (tests only): the ranks exchange through the host and may share a GPU, so the whole N > 1
    # control flow of this file (every rank's collectives in the same order, rank 0's JSON line) runs on a 1-GPU box
    backend = os.environ.get("SAIS_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if dist_on:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        world = dist.get_world_size()                    # what RCCL actually formed

    import synth
    from sais_amd import ops
    B, T, C = args.clips, args.frames, 2
    two = args.two_stream
    vit, model, protos, opt = build(dev, B, T, C, lr=0.1, two_stream=two, drop_path=args.vit_drop_path)
    frames = synth.clips(seed=rank, B=B, T=T).view(B * T, 3, 224, 224).to(dev)     # resident in HBM
    if two:      # the flow stream: seeded synthetic frames of the same shape (RAFT renders flow as RGB images, :62-67)
        frames = torch.cat([frames, synth.clips(seed=1000 + rank, B=B, T=T).view(B * T, 3, 224, 224).to(dev)])
    pad = synth.padding_mask([T] * B).to(dev)
    labels = synth.labels(seed=rank, B=B, nclasses=C)
    from sais_amd.parallel import GradSync
    sync = GradSync(world, active=dist_on, payload_dtype=torch.bfloat16 if args.grad_payload == "bf16" else None)
    comm_events = CommEvents() if dist_on else None
    step = make_step(vit, model, protos, opt, sync, frames, pad, labels, B, T, world, dist_on, comm_events, two_stream=two)
    vit(frames[:2])                                          # builds the flat buffers
    model._engine(dev)
    vit.grad_ready_hook = sync.vit_hook(vit)
    model.grad_ready_hook = sync.temporal_hook(model, T)

    eager_step = step
    # ONE launch path for every N: the whole step — kernels AND, with N > 1, the per-block RCCL all-reduces that the
    # backward hooks issue on ProcessGroupNCCL's stream (forked / joined with events, which the capture records as graph
    # edges) — is captured into one hipGraph.  gloo (tests on a 1-GPU box) exchanges through the host: not capturable.
    use_graph = not args.no_graph and (not dist_on or backend == "nccl")
    graph_note = None
    if use_graph:
        from sais_amd.graph import GraphedStep
        try:
            step = GraphedStep(eager_step, warmup=2, capture_error_mode="thread_local" if dist_on else "global")
        except Exception as e:                               # noqa: BLE001
            if not dist_on:
                raise
            # N > 1 only: the RCCL build refused the capture.  Every rank takes this branch together (the capture is
            # collective), the run continues on the eager launch path and SAYS so in the line and on stderr.
            graph_note = f"hipGraph capture of the RCCL step failed ({type(e).__name__}: {e}); eager launch path"
            sys.stderr.write("bench.py: " + graph_note + "\n")
            use_graph = False
            torch.cuda.synchronize()
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
    timed_loss = float(loss.detach())

    # parity of the measured code path: one hipGraph replay and one eager step from the SAME weights and the same dropout
    # RNG state must give the same loss (the forward has no atomics, so this is exact up to nothing)
    graph_check = None
    snap = [vit.flat.flat.clone(), model.flat.flat.clone()] + [p.detach().clone() for p in protos.values()]
    rng_snap = None if model._rng is None else model._rng.clone()       # train-mode dropout: same masks for both runs
    dp_snap = None if vit._rng is None else vit._rng.clone()            # ... and the same DropPath draws

    def restore():
        with torch.no_grad():
            if rng_snap is not None:
                model._rng.copy_(rng_snap)
            if dp_snap is not None:
                vit._rng.copy_(dp_snap)
            vit.flat.flat.copy_(snap[0])
            model.flat.flat.copy_(snap[1])
            for p, s in zip(protos.values(), snap[2:]):
                p.copy_(s)
        vit.flat.refresh_shadows(vit._t_names)
        model.flat.refresh_shadows(model._t_names())
    if use_graph and (rank == 0 or dist_on):      # N > 1: a step contains collectives, so every rank runs the check
        lg = float(step().detach())
        restore()
        le = float(eager_step().detach())
        restore()
        graph_check = dict(loss_graph_replay=lg, loss_eager=le, abs_diff=abs(lg - le))
        if not abs(lg - le) <= 1e-6 * max(1.0, abs(le)):
            if world == 1:
                raise SystemExit(f"bench.py: hipGraph replay loss {lg!r} != eager loss {le!r} from the same weights")
            # N > 1: one rank leaving would strand the others in their next collective; the line carries the mismatch
            graph_check["mismatch"] = True
            sys.stderr.write(f"bench.py[rank {rank}]: hipGraph replay loss {lg!r} != eager loss {le!r}\n")

    # in-run parity gate (rank 0): the timed model on the timed inputs vs the CPU oracle, same draws
    parity = None
    if rank == 0 and args.parity_clips > 0:
        restore()
        parity = parity_gate(vit, model, protos, frames, pad, labels, B, T, C, two, args.parity_clips)
        restore()
        if not parity["pass"]:
            sys.stderr.write(f"bench.py: PARITY GATE FAILED: max-abs logit deviation {parity['max_abs_logit']} > 1e-3\n")

    # instrumented pass (outside the timed region): HIP events around every MFMA kernel launch
    # With N > 1 a step contains collectives (gradient all-reduces from the backward hooks), so EVERY rank runs these
    # passes; only rank 0 brackets its kernels with events.
    roof = None
    NPASS = 3
    if rank == 0:
        ops.TIMER = ops.KernelTimer()
    if comm_events is not None:
        comm_events.enabled = True
    if rank == 0 or dist_on:
        for _ in range(NPASS):
            eager_step()
        torch.cuda.synchronize()
    comm = None
    if comm_events is not None and comm_events.recs:
        r = comm_events.recs
        comm = dict(allreduce_bytes_per_step=int(r[-1][2]), allreduce_launches_per_step=int(r[-1][3]),
                    exposed_comm_ms_per_step=round(sum(a.elapsed_time(b) for a, b, _, _ in r) / len(r), 3),
                    exposed_comm_measured_in="eager instrumented passes after the timed region (HIP events around the "
                                             "join; events cannot be timed inside a captured graph)",
                    payload_dtype=args.grad_payload,
                    buckets=[dict(kind=k, mbytes=round(b / 2 ** 20, 2), slices=n) for k, b, n in sync.last_buckets],
                    bucket_rule=f">= {sync.bucket_bytes >> 20} MiB of adjacent flat-gradient slices per all-reduce, in "
                                f"backward (reverse-layer) order; slices < {sync.small_bytes >> 20} MiB packed into one",
                    payload="flat gradient slices handed over by the backward hooks (temporal encoder, then ViT blocks "
                            "last..first) and coalesced into buckets, captured into the step's hipGraph"
                            + ("" if use_graph else " (eager launch path)"))
    # tests only (SAIS_BENCH_RANK_DUMP=<dir>): every rank leaves its own view of the exchange — the bucket plan must be the
    # same list on every rank (collectives are matched by order), the data must differ (rank-seeded), the loss is its own
    dump_dir = os.environ.get("SAIS_BENCH_RANK_DUMP")
    if dump_dir:
        with open(os.path.join(dump_dir, f"rank{rank}.json"), "w") as fh:
            json.dump({"rank": rank, "world": world, "buckets": [list(b) for b in sync.last_buckets],
                       "frames_checksum": float(frames.double().sum()), "loss": timed_loss,
                       "payload": args.grad_payload}, fh)
    if rank == 0:
        summ = ops.TIMER.summary()
        ops.TIMER = None
        # the dominant kernel = the (kernel, shape) tag with the largest total time; a kernel that runs several shapes has
        # one launch duration per shape, so the shapes are not pooled
        kern = max(summ, key=lambda t: summ[t]["total_ms"])
        k = summ[kern]
        avg_ms = k["avg_ms"]
        flops, nbytes = k["flops"], k["bytes"]
        ach = flops / (avg_ms * 1e-3) / 1e12
        # key of the same kernel in profiles/pmc_traffic.json ("gemm_ln_bwd[N384,K1536]" -> "gemm_ln_bwd[K1536]")
        base = kern.split("[")[0]
        pmc_key = base + "[" + kern.split(",")[-1] if base in ("gemm_ln_fwd", "gemm_ln_bwd") else base
        if base == "gemm_tn_grouped" and "GEMMs" in kern:
            # the dW launches of a step share one kernel and differ in size: the counter file keys them by workgroup count.  ViT-S: 24
            # tiles of 192 x 384 per block (4 GEMMs), 4 for the k / v gradient of the CLS-only block; splits = 256 // tiles
            n_gemm = int(kern.split("[")[1].split(" ")[0])
            tiles = (n_gemm // 4) * 24 + (n_gemm % 4) * 4
            pmc_key = "gemm_tn_grouped[%d wg]" % (tiles * max(1, 256 // tiles))
        pmc_all, pmc_stale = load_pmc("pmc_traffic.json")
        pmc = None if pmc_stale else (pmc_all or {}).get(pmc_key)
        roof = dict(bound="mfma", kernel=kern, achieved=round(ach, 1), peak=MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                    frac=round(ach / MFMA_PEAK_TFLOPS, 4),
                    traffic=(pmc or {}).get("hbm_bytes_per_launch"),
                    traffic_stale=bool(pmc_stale),
                    traffic_source="profiles/pmc_traffic.json (rocprofv3 --pmc passes of this command, collected "
                                   "offline; NOT a live counter; null + traffic_stale when the kernel sources have "
                                   "changed since: the file carries their hash)",
                    traffic_collected=(pmc_all or {}).get("_provenance"),
                    algorithmic_bytes=int(nbytes),
                    hbm_gbps_algorithmic=round(nbytes / (avg_ms * 1e-3) / 1e9, 1),
                    avg_launch_us=round(avg_ms * 1e3, 1), launches_per_step=k["launches"] // NPASS,
                    hbm_frac_of_this_kernel=round(nbytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                    timing="raw HIP-event interval per launch (includes ~4 us of event overhead; rocprofv3's kernel "
                           "durations in profiles/ are shorter by that much)",
                    all_kernels={n: dict(ms_per_step=round(v["total_ms"] / NPASS, 3), launches_per_step=v["launches"] // NPASS,
                                         avg_us=round(v["avg_ms"] * 1e3, 1),
                                         tflops=round(v["flops"] / (v["avg_ms"] * 1e-3) / 1e12, 1),
                                         gbps=round(v["bytes"] / (v["avg_ms"] * 1e-3) / 1e9, 1)) for n, v in summ.items()})
    # sustained rate (outside the metric): the timed region is a fraction of a second, so keep running the same step
    # for >= --sustain-seconds and report what the chip settles at under sustained MFMA load.  The weights are put back to
    # the snapshot before every chunk (untimed): a thousand SGD steps on random labels would otherwise drift to values
    # whose arithmetic no longer costs what real data costs.
    sustained = None
    if args.sustain_seconds > 0:
        n_chunk = max(args.steps, 10)
        ts0 = time.perf_counter()
        last_dt, total = 0.0, 0
        while True:
            restore()
            torch.cuda.synchronize()
            tc = time.perf_counter()
            for _ in range(n_chunk):
                step()
            torch.cuda.synchronize()
            last_dt = time.perf_counter() - tc
            total += n_chunk
            left = torch.tensor([args.sustain_seconds - (time.perf_counter() - ts0)], device=dev)
            if dist_on:                                       # every rank leaves the loop in the same iteration
                dist.all_reduce(left, op=dist.ReduceOp.MIN)
            if left.item() <= 0:
                break
        tl = torch.tensor([last_dt], device=dev)
        if dist_on:
            dist.all_reduce(tl, op=dist.ReduceOp.MAX)
        sustained = dict(seconds=round(time.perf_counter() - ts0, 1), steps=total,
                         frames_per_s=round(world * B * T * n_chunk / tl.item(), 1),
                         note="rate of the LAST %d steps of a >= %.0f s back-to-back run of the same step" %
                              (n_chunk, args.sustain_seconds))
    if dist_on:
        dist.barrier()

    if rank == 0:
        fps = world * B * T * args.steps / dt
        nstream = 2 if two else 1
        pruned = bool(getattr(vit, "prune_last_block", False))
        from sais_amd import vit as _vitmod
        per_frame = FLOP_PER_FRAME_FWD_BWD - (FLOP_PRUNED_PER_FRAME + (FLOP_PRUNED_Q_PER_FRAME if _vitmod._PRUNE_Q else 0.0)
                                              if pruned else 0.0)
        step_flops = nstream * (B * T * per_frame + B * FLOP_TEMPORAL_PER_CLIP)
        if roof is not None:
            roof["hbm_frac"] = round(nstream * B * T * HBM_BYTES_PER_FRAME * args.steps / dt / 1e9 / HBM_PEAK_GBPS, 4)
            roof["hbm_frac_note"] = "MODEL-based: 133 MB/frame fused-plan algorithmic bytes / step time / 8 TB/s"
            step_pmc = None if pmc_stale else (pmc_all or {}).get("_step", {}).get("step_hbm_bytes")
            if step_pmc and not two and (B, T) == (8, 32):
                roof["hbm_frac_counters"] = round(step_pmc * args.steps / dt / 1e9 / HBM_PEAK_GBPS, 4)
                roof["hbm_frac_counters_note"] = ("COUNTER-based: sum over kernel families of launches x PMC bytes per "
                                                  "launch (profiles/pmc_traffic.json, collected offline on this config) "
                                                  "/ this run's step time / 8 TB/s")
        out = {
            "metric": "frames/sec ViT-S/16 fwd+bwd, 224x224 32-frame clips",
            "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": (f"BASELINE config 4 (two-stream): RGB + optical-flow frames, 2 x {B} x {T} frames through "
                                    f"ViT-S/16, 4-layer temporal encoder per stream, fused by add, SupCon prototype loss, "
                                    f"fwd+bwd+SGD; value counts the {B * T} RGB frames per step" if two else
                                    f"BASELINE config 2: ViT-S/16 + 4-layer temporal encoder + SupCon prototype loss, "
                                    f"fwd+bwd+SGD, {B} clips x {T} frames x 224x224 per GPU (global {world * B} clips), "
                                    f"random-init weights, RGB stream"), "clips_per_gpu": B, "frames_per_clip": T,
                       "streams": nstream, "vit_frames_per_step_per_gpu": nstream * B * T,
                       "parallelism": f"dp{world}", "launch": "hipGraph replay" if use_graph else (graph_note or "eager"),
                       "temporal_dropout": model.dropout_p, "vit_drop_path": vit.drop_path_rate,
                       "last_block_cls_only": pruned,
                       "flops_per_frame": {"reference_dense": FLOP_PER_FRAME_FWD_BWD, "executed": per_frame,
                                           "note": "forward() returns the CLS row: the last block's row-local half and the "
                                                   "non-CLS queries of its attention feed nothing and are not computed; "
                                                   "outputs and all parameter gradients are unchanged (tests vs the oracle). "
                                                   "SAIS_VIT_PRUNE_LAST=0 computes every row; step_tflops and "
                                                   "frac_of_mfma_roofline use the executed count"}},
            "step_tflops": round(step_flops * world * args.steps / dt / 1e12, 1),
            "frac_of_mfma_roofline": round(step_flops * args.steps / dt / 1e12 / MFMA_PEAK_TFLOPS, 4),
            "loss": round(timed_loss, 6),
            "sustained": None if sustained is None else dict(sustained, ratio_to_value=round(sustained["frames_per_s"] / fps, 4)),
            "graph_vs_eager": graph_check,
            "parity": parity,
            "comm": comm,
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                avail = len(os.sched_getaffinity(0))
            except AttributeError:
                avail = os.cpu_count() or 1
            # fp32 ViT-S on one 32-frame clip stops scaling past a few dozen threads (and oversubscribed
            # hosts get much slower), so use at most 32 of the host's cores; `cores` reports what was used
            weights = ({k: v.detach().float().cpu() for k, v in vit.state_dict().items()},
                       {k: v.detach().float().cpu() for k, v in model.state_dict().items()
                        if not k.startswith(("clip_", "transEncoderClip", "attention", "finalModules", "linear2"))},
                       {k: v.detach().float().cpu() for k, v in protos.items()})
            out["cpu_baseline"] = cpu_baseline(T, C, max(1, min(avail, 32)), weights, Bfull=B)
        if world == 1 and not dist_on and not two and not args.no_variants and os.environ.get("SAIS_BENCH_BACKEND", "nccl") == "nccl":
            out["variants"] = run_variants(args)
        # a number from a numerically wrong step is not a result: the line says so and the exit code is non-zero
        bad = (parity is not None and not parity["pass"]) or bool(graph_check and graph_check.get("mismatch"))
        if bad:
            out["invalid"] = True
        print(json.dumps(out), flush=True)
    else:
        bad = False
    if dist_on:
        flag = torch.tensor([1 if bad else 0], device=dev)
        dist.broadcast(flag, src=0)                       # every rank leaves with rank 0's verdict
        bad = bool(flag.item())
        dist.destroy_process_group()
    if bad:
        sys.exit(3)


if __name__ == "__main__":
    main()
