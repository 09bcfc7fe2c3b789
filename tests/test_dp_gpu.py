"""Data-parallel path on real GPUs: N ranks x B clips must produce, after the gradient all-reduce, the
gradient one process computes on the concatenated N*B-clip batch (the loss is a mean over clips, the exchange is
sum / world).  Two launch forms:

  * 2 ranks over RCCL on 2 GPUs                    — skipped on a 1-GPU box;
  * 2 ranks sharing cuda:0 over gloo               — runs on a 1-GPU box: same GradSync hooks, same flat-gradient
                                                      slices, only the transport differs (RCCL refuses two ranks on
                                                      one device);
plus `bench.py`'s launcher contract: `--gpus N` starts N ranks or fails loudly, and the distributed branch of bench.py
(RCCL init, hook all-reduces, barriers, max-over-ranks timing) runs end to end with world size 1.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, T, DEPTH = 2, 8, 2


def _build(dev):
    from sais_amd.temporal import fullModel
    from sais_amd.vit import vit_small
    vit = vit_small(patch_size=16, depth=DEPTH)
    vit.load_state_dict(synth.vit_state_dict(seed=0, depth=DEPTH), strict=True)
    m = fullModel('reps', 2, 'in_vs_out', 384, 'ViT', modalities='RGB')
    m.load_state_dict(synth.temporal_state_dict(seed=1), strict=True)
    m.dropout_p = 0.0                      # ranks would draw different masks than the single-process run
    protos = torch.nn.ParameterDict({k: torch.nn.Parameter(v.clone().to(dev)) for k, v in synth.prototypes(2, 2).items()})
    return vit.to(dev).train(), m.to(dev).train(), protos


def _grads(vit, m, protos, clips, lens, labels, dev, sync=None):
    from sais_amd.loss import calcNCELoss
    n = clips.shape[0]
    frames = clips.view(n * T, 3, 224, 224).to(dev)
    pad = synth.padding_mask(lens, maxT=T).to(dev)
    vit(frames[:1])
    m._engine(dev)
    if sync is not None:
        vit.grad_ready_hook = sync.vit_hook(vit)
        m.grad_ready_hook = sync.temporal_hook(m)
    vit.flat.grad.zero_()
    m.flat.grad.zero_()
    reps = vit(frames).view(n, 1, T, 384)
    emb, _ = m(reps, None, lens, None, 'Prototypes', pad, None, None)
    loss = calcNCELoss(0, emb, labels, ["v"] * n, protos, None)
    loss.backward()
    if sync is not None:
        sync.reduce_params(protos.values())
        sync.wait()
    torch.cuda.synchronize()
    return vit.flat.grad.clone(), m.flat.grad.clone(), [p.grad.clone() for p in protos.values()]


def _inputs(world):
    clips = synth.clips(seed=77, B=world * B, T=T)
    lens = [T - (b % 3) for b in range(world * B)]
    labels = synth.labels(seed=78, B=world * B)
    return clips, lens, labels


def _worker(rank, world, port, backend, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from sais_amd.parallel import GradSync
    vit, m, protos = _build(dev)
    clips, lens, labels = _inputs(world)
    sl = slice(rank * B, (rank + 1) * B)
    sync = GradSync(world)
    gv, gm, gp = _grads(vit, m, protos, clips[sl], lens[sl], labels[sl], dev, sync)
    if rank == 0:
        torch.save(dict(gv=gv.cpu() / world, gm=gm.cpu() / world, gp=[g.cpu() / world for g in gp]), out)
    dist.barrier()
    dist.destroy_process_group()


def _compare(out, world):
    dev = torch.device("cuda", 0)
    vit, m, protos = _build(dev)
    gv, gm, gp = _grads(vit, m, protos, *_inputs(world), dev)
    got = torch.load(out)

    def rel(a, b):
        return float((a - b).norm() / b.norm().clamp_min(1e-12))
    # same kernels on both sides; what differs is the summation order (per-rank partial sums + all-reduce vs one
    # longer M loop with fp32 atomics) and bf16 rounding of per-rank activations being identical -> tight bound
    assert rel(got["gv"], gv.cpu()) <= 2e-3, rel(got["gv"], gv.cpu())
    from sais_amd.parallel import GradSync
    touched = torch.zeros(m.flat.numel, dtype=torch.bool)
    for a, b in GradSync.temporal_ranges(m, T):
        touched[a:b] = True
    assert rel(got["gm"][touched], gm.cpu()[touched]) <= 2e-3
    assert float(gm.cpu()[~touched].abs().max()) == 0.0       # nothing outside the exchanged slices has a gradient
    for a, b in zip(got["gp"], gp):
        assert rel(a, b.cpu()) <= 2e-3


def _spawn(world, backend, tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "dp.pt")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, backend, out), nprocs=world, join=True)
    return out


def test_two_ranks_rccl_gradients_equal_single_process(tmp_path):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (the driver's multi-GPU node); the gloo variant below runs on one")
    _compare(_spawn(2, "nccl", tmp_path), 2)


def test_two_ranks_on_one_gpu_gloo_gradients_equal_single_process(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _compare(_spawn(2, "gloo", tmp_path), 2)


def test_bench_gpus_flag_starts_n_ranks_or_fails_loudly():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    n = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, cwd=ROOT)
    assert r.returncode != 0 and "refusing" in r.stderr and not r.stdout.strip()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], capture_output=True, text=True,
                       cwd=ROOT, env=dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
    if n >= 2:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                            "--no-cpu-baseline", "--sustain-seconds", "0"], capture_output=True, text=True, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert line["n_gpus"] == 2 and line["comm"]["allreduce_bytes_per_step"] > 100e6


def _force_dist_bench(*extra):
    port = 29600 + (os.getpid() % 2000)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--sustain-seconds", "0",
                        *extra],
                       capture_output=True, text=True, cwd=ROOT, timeout=900,
                       env=dict(os.environ, SAIS_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_distributed_branch_with_world_size_one():
    """SAIS_BENCH_FORCE_DIST=1 under torchrun --nproc-per-node 1: RCCL process group, hook-driven all-reduces of
    every gradient slice (~122 MB per step) CAPTURED INTO THE STEP'S hipGraph (the same launch path as N = 1), barriers
    and the rank-0 JSON line; the replayed graph and an eager step from the same weights must give the same loss."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    line = _force_dist_bench()
    assert line["n_gpus"] == 1 and line["config"]["launch"] == "hipGraph replay"
    assert line["graph_vs_eager"]["abs_diff"] <= 1e-6
    assert 100e6 < line["comm"]["allreduce_bytes_per_step"] < 140e6       # 30.45 M touched params x 4 B
    assert line["comm"]["exposed_comm_ms_per_step"] >= 0 and line["comm"]["payload_dtype"] == "fp32"
    # round 4: the ~20 per-slice all-reduces are coalesced (sais_amd.parallel.GradSync): the temporal encoder's slice, four
    # buckets of three ViT blocks (>= 16 MiB each) and ONE packed exchange of every small slice
    assert line["comm"]["allreduce_launches_per_step"] <= 8
    b = line["comm"]["buckets"]
    assert len(b) == line["comm"]["allreduce_launches_per_step"] and b[-1]["kind"] == "packed"
    assert all(x["mbytes"] >= 16 for x in b if x["kind"] == "bucket") and [x["slices"] for x in b[:-1]] == [1, 3, 3, 3, 3]


def test_bench_distributed_branch_eager_launch_path():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    line = _force_dist_bench("--no-graph")
    assert line["config"]["launch"] == "eager" and 100e6 < line["comm"]["allreduce_bytes_per_step"] < 140e6


def test_bench_two_ranks_complete_and_print_one_line(tmp_path):
    """The N > 1 control flow of bench.py end to end on this 1-GPU box: two ranks under torch.distributed.run, exchanging
    through gloo (SAIS_BENCH_BACKEND, tests only) and sharing the GPU — every rank must issue the same collectives in the
    same order (a rank-0-only eager pass with all-reduces in it would hang here), rank 0 prints the one JSON line."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import json
    env = dict(os.environ, SAIS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", str(29700 + os.getpid() % 200), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--clips", "2", "--frames", "8",
                        "--sustain-seconds", "0"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["scaling"] == "weak"
    assert d["comm"]["allreduce_bytes_per_step"] > 80e6 and d["value"] > 0 and d["roofline"]["all_kernels"]


def test_bench_eight_ranks_over_gloo_on_one_gpu(tmp_path):
    """World 8 without a second GPU: bench.py's distributed branch with eight ranks under torch.distributed.run,
    exchanging through gloo and sharing cuda:0 (1 clip x 8 frames per rank, bf16 gradient payload).  Every rank must build
    the SAME bucket list (collectives are matched by issue order: one rank with a different plan would hang or corrupt),
    the ranks must hold DIFFERENT data (rank-seeded), and rank 0 prints exactly one line with n_gpus 8."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import json
    dump = tmp_path / "ranks"
    dump.mkdir()
    env = dict(os.environ, SAIS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", SAIS_BENCH_RANK_DUMP=str(dump))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr",
                        "127.0.0.1", "--master-port", str(29900 + os.getpid() % 90), os.path.join(ROOT, "bench.py"),
                        "--gpus", "8", "--steps", "2", "--warmup", "1", "--clips", "1", "--frames", "8",
                        "--sustain-seconds", "0", "--parity-clips", "0", "--grad-payload", "bf16"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["parallelism"] == "dp8" and d["scaling"] == "weak" and d["value"] > 0
    assert d["comm"]["payload_dtype"] == "bf16" and 40e6 < d["comm"]["allreduce_bytes_per_step"] < 70e6      # half of ~122 MB
    ranks = [json.load(open(dump / f"rank{i}.json")) for i in range(8)]
    assert all(x["world"] == 8 for x in ranks)
    assert all(x["buckets"] == ranks[0]["buckets"] for x in ranks) and len(ranks[0]["buckets"]) >= 3
    assert len({x["frames_checksum"] for x in ranks}) == 8, "every rank must draw its own clips"
