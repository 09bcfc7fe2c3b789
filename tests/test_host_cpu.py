"""CPU-only host logic: state_dict contracts, checkpoint codec, flat parameter storage, the 2-rank
gradient exchange (gloo), and the loud failure without a device."""
import os

import pytest
import torch
import torch.nn as nn

import synth


def test_state_dict_contracts_match_reference_key_lists():
    from sais_amd.temporal import fullModel
    from sais_amd.vit import vit_small
    v = vit_small(patch_size=16, drop_path_rate=0.1)
    assert list(v.state_dict().keys()) == [k for k, _, _ in synth.vit_keys()]
    assert sum(p.numel() for p in v.parameters()) == 21665664                 # SURVEY §8c probe
    m = fullModel('reps', 2, 'in_vs_out', 384, 'ViT', modalities='RGB-Flow')
    want = {k: s for k, s, _ in synth.temporal_keys()}
    got = {k: tuple(t.shape) for k, t in m.state_dict().items()}
    assert got == want and len(got) == 4118
    assert sum(p.numel() for p in m.parameters()) == 19180681                 # SURVEY App. A probe
    keys = list(m.state_dict().keys())
    i0 = keys.index("frame_pos_embeddings.0")
    assert keys[i0:i0 + 12] == [f"frame_pos_embeddings.{i}" for i in range(12)]   # insertion order, not sorted


def test_out_of_scope_branches_raise():
    from sais_amd.temporal import fullModel
    with pytest.raises(NotImplementedError):
        fullModel('raw', 2, 'd', 512, 'R3D')
    with pytest.raises(NotImplementedError):
        fullModel('reps', 2, 'd', 384, 'ViT', modalities='Flow', importance_loss=True)
    mi = fullModel('reps', 2, 'd', 384, 'ViT', importance_loss=True)           # -il adds importance_function.{weight,bias}
    assert {k: tuple(t.shape) for k, t in mi.state_dict().items()} == {k: s for k, s, _ in synth.temporal_keys(importance=True)}
    m = fullModel('reps', 2, 'd', 384, 'ViT')
    with pytest.raises(NotImplementedError):
        m(None, None, None, None, 'MIL', None, None, None)


def test_no_cpu_fallback():
    from sais_amd._lib import SaisHipError
    from sais_amd.loss import calcNCELoss
    from sais_amd.temporal import fullModel
    from sais_amd.vit import vit_small
    with pytest.raises(SaisHipError):
        vit_small()(torch.zeros(1, 3, 224, 224))
    m = fullModel('reps', 2, 'd', 384, 'ViT', modalities='RGB')
    with pytest.raises(SaisHipError):
        m(torch.zeros(1, 1, 4, 384), None, [4], None, 'Prototypes', None, None, None)
    with pytest.raises(SaisHipError):
        calcNCELoss(0, torch.zeros(2, 256), torch.tensor([0, 1]), ["a", "b"], synth.prototypes(2, 2), None)


def test_checkpoint_codec_roundtrip(tmp_path):
    from sais_amd import model_io
    from sais_amd.temporal import fullModel
    m = fullModel('reps', 3, 'in_vs_out', 384, 'ViT')
    m.load_state_dict(synth.temporal_state_dict(seed=7))
    protos = nn.ParameterDict({k: nn.Parameter(v) for k, v in synth.prototypes(3, 3).items()})
    model_io.save_params_file(m, tmp_path / "params.zip")
    model_io.save_prototypes_file(protos, tmp_path / "prototypes.zip")
    raw = torch.load(tmp_path / "params.zip", weights_only=False)
    assert all(k.startswith("module.") for k in raw) and len(raw) == 4118
    raw["module.encoder.cls_token"] = torch.zeros(1, 1, 768)                   # timm ballast must be ignored
    torch.save(raw, tmp_path / "params.zip")
    md, opt, dev = model_io.loadModel(0, 1, str(tmp_path), 'reps', 3, 'in_vs_out', 384, 'ViT', 'Prototypes', 0,
                                      lr=0.1, modalities='RGB-Flow', inference=True, device='cpu')
    sd = md['model'].state_dict()
    for k, v in synth.temporal_state_dict(seed=7).items():
        assert torch.equal(sd[k], v), k
    assert list(md['prototypes'].keys()) == ['0', '1', '2']
    assert torch.equal(md['prototypes']['2'].detach(), synth.prototypes(3, 3)['2'])
    # a key without the DDP prefix fails like the reference's split('module.')[1]
    torch.save({"frame_cls": torch.zeros(1, 384)}, tmp_path / "params.zip")
    with pytest.raises(IndexError):
        model_io.load_params_file(tmp_path / "params.zip")
    # strict: a missing key raises
    bad = {"module." + k: v for k, v in synth.temporal_state_dict(seed=7).items() if k != "linear.bias"}
    torch.save(bad, tmp_path / "params.zip")
    with pytest.raises(RuntimeError):
        model_io.loadModel(0, 1, str(tmp_path), 'reps', 3, 'in_vs_out', 384, 'ViT', 'Prototypes', 0,
                           inference=True, device='cpu')


def test_flat_params_views_and_grad_attach():
    from sais_amd.flat import FlatParams
    from sais_amd.vit import vit_small
    v = vit_small(depth=2)
    ref = {k: t.clone() for k, t in v.state_dict().items()}
    f = FlatParams(v, 'cpu')
    assert f.intact() and f.numel % 4 == 0
    for k, t in v.state_dict().items():
        assert torch.equal(t, ref[k])
    f.flat.mul_(2.0)                                   # parameters are views of the flat buffer
    assert torch.equal(v.blocks[1].mlp.fc2.weight.detach(), 2 * ref["blocks.1.mlp.fc2.weight"])
    f.g("norm.bias").fill_(3.0)                        # p.grad are views of the flat gradient buffer
    assert float(v.norm.bias.grad.sum()) == 3.0 * 384
    for p in v.parameters():
        p.grad = None                                  # optimizer.zero_grad(set_to_none=True)
    assert f.attach_grads() is True and float(f.grad.abs().sum()) == 0.0
    assert v.norm.bias.grad.data_ptr() == f.g("norm.bias").data_ptr()
    lo, hi = 0, 0
    v.flat = f
    spans = [v.block_grad_range(i) for i in range(2)]
    assert spans[0][1] == spans[1][0] and spans[1][1] == f.offsets["norm.weight"]


def _dp_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sais_amd.flat import FlatParams
    from sais_amd.parallel import GradSync
    from sais_amd.temporal import fullModel
    from sais_amd.vit import vit_small
    torch.manual_seed(0)
    vit, m = vit_small(depth=2), fullModel('reps', 2, 'd', 384, 'ViT', modalities='RGB')
    vit.flat, m.flat = FlatParams(vit, 'cpu'), FlatParams(m, 'cpu')
    T = 5
    sync = GradSync(world)
    g = torch.Generator().manual_seed(100 + rank)
    vit.flat.grad.copy_(torch.randn(vit.flat.numel, generator=g))
    m.flat.grad.copy_(torch.randn(m.flat.numel, generator=g))
    local_v, local_m = vit.flat.grad.clone(), m.flat.grad.clone()
    # backward order: temporal first, then final norm, blocks last..first, embedding
    # (the temporal hook fires once per backward CALL — three times with TTA list inputs — but the slices must be
    # exchanged ONCE, over the longest stream seen: a second all-reduce would double-count the remote gradients)
    m._touched_T = T
    th = sync.temporal_hook(m, T)
    th(0, m.flat.numel)
    th(0, m.flat.numel)
    hook = sync.vit_hook(vit)
    hook(vit.flat.offsets["norm.weight"], vit.flat.numel)
    for i in reversed(range(2)):
        hook(*vit.block_grad_range(i))
    hook(0, vit.flat.offsets["blocks.0.norm1.weight"])
    nbytes = sync.wait()
    others_v = [torch.zeros_like(local_v) for _ in range(world)]
    others_m = [torch.zeros_like(local_m) for _ in range(world)]
    dist.all_gather(others_v, local_v)
    dist.all_gather(others_m, local_m)
    ok_v = torch.allclose(vit.flat.grad, sum(others_v), atol=1e-6)
    touched = torch.zeros(m.flat.numel, dtype=torch.bool)
    for a, b in GradSync.temporal_ranges(m, T):
        touched[a:b] = True
    ok_m = torch.allclose(m.flat.grad[touched], sum(others_m)[touched], atol=1e-6) and \
        torch.equal(m.flat.grad[~touched], local_m[~touched])
    # every parameter the Prototypes path touches lies inside the exchanged ranges
    names_touched = ["frame_cls", "linear.weight", "linear.bias", "frame_pos_embeddings.4",
                     "transEncoderFrame.layers.3.norm2.bias", "transEncoderFrame.layers.0.self_attn.in_proj_weight"]
    cover = all(touched[m.flat.offsets[n]] for n in names_touched) and not touched[m.flat.offsets["frame_pos_embeddings.5"]] \
        and not touched[m.flat.offsets["transEncoderClip.layers.0.linear1.weight"]]
    if rank == 0:
        open(out, "w").write(f"{ok_v} {ok_m} {cover} {nbytes}")
    dist.destroy_process_group()


def test_data_parallel_gradient_exchange_gloo_world2(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "res.txt")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_dp_worker, args=(2, port, out), nprocs=2, join=True)
    ok_v, ok_m, cover, nbytes = open(out).read().split()
    assert (ok_v, ok_m, cover) == ("True", "True", "True")
    assert int(nbytes) > 4 * 3_000_000


def _bucket_worker(rank, world, port, out):
    import json
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sais_amd.flat import FlatParams
    from sais_amd.parallel import GradSync
    from sais_amd.temporal import fullModel
    from sais_amd.vit import vit_small
    torch.manual_seed(0)
    vit, m = vit_small(depth=12), fullModel('reps', 2, 'd', 384, 'ViT', modalities='RGB')
    vit.flat, m.flat = FlatParams(vit, 'cpu'), FlatParams(m, 'cpu')
    T = 32
    protos = [torch.nn.Parameter(torch.zeros(1, 256)) for _ in range(2)]
    g = torch.Generator().manual_seed(200 + rank)
    vit.flat.grad.copy_(torch.randn(vit.flat.numel, generator=g))
    m.flat.grad.copy_(torch.randn(m.flat.numel, generator=g))
    for q in protos:
        q.grad = torch.randn(1, 256, generator=g)
    local = [vit.flat.grad.clone(), m.flat.grad.clone()] + [q.grad.clone() for q in protos]
    sync = GradSync(world)
    logs = []
    for step in range(2):                              # two steps: the bucket plan is the same every step
        if step:
            vit.flat.grad.copy_(local[0]); m.flat.grad.copy_(local[1])
            for q, l in zip(protos, local[2:]):
                q.grad.copy_(l)
        m._touched_T = T
        sync.temporal_hook(m, T)(0, m.flat.numel)
        hook = sync.vit_hook(vit)                       # the order of VisionTransformer._backward_kernels
        hook(vit.flat.offsets["norm.weight"], vit.flat.numel)
        for i in reversed(range(12)):
            hook(*vit.block_grad_range(i))
        hook(0, vit.flat.offsets["blocks.0.norm1.weight"])
        sync.reduce_params(protos)
        nbytes = sync.wait()
        logs.append([list(b) for b in sync.last_buckets])
    gathered = []
    for l in local:
        parts = [torch.zeros_like(l) for _ in range(world)]
        dist.all_gather(parts, l)
        gathered.append(sum(parts))
    touched = torch.zeros(m.flat.numel, dtype=torch.bool)
    for a, b in GradSync.temporal_ranges(m, T):
        touched[a:b] = True
    ok = torch.allclose(vit.flat.grad, gathered[0], atol=1e-6) \
        and torch.allclose(m.flat.grad[touched], gathered[1][touched], atol=1e-6) \
        and torch.equal(m.flat.grad[~touched], local[1][~touched]) \
        and all(torch.allclose(q.grad, s, atol=1e-6) for q, s in zip(protos, gathered[2:]))
    json.dump(dict(ok=bool(ok), logs=logs, nbytes=nbytes), open(out + f".{rank}", "w"))
    dist.destroy_process_group()


def test_gradient_buckets_are_identical_on_both_ranks_and_at_least_16_mib_gloo_world2(tmp_path):
    """the ~20 per-slice all-reduces of a config-2 step are coalesced into <= 8 collectives (>= 16 MiB of
    adjacent slices each, small slices packed into one), the same boundaries in the same order on every rank and every
    step, and the exchanged gradients equal the sum over the ranks."""
    import json
    import torch.multiprocessing as mp
    out = str(tmp_path / "buckets")
    mp.spawn(_bucket_worker, args=(2, 30500 + (os.getpid() % 2000), out), nprocs=2, join=True)
    r0, r1 = (json.load(open(out + f".{r}")) for r in range(2))
    assert r0["ok"] and r1["ok"]
    assert r0["logs"] == r1["logs"] and r0["logs"][0] == r0["logs"][1]
    plan = r0["logs"][0]
    assert len(plan) <= 8, plan
    buckets = [b for b in plan if b[0] == "bucket"]
    assert all(b[1] >= 16 << 20 for b in buckets), plan                    # every bucket reached the threshold
    assert [b[0] for b in plan].count("packed") == 1 and plan[-1][0] == "packed"
    # reverse-layer order: the temporal encoder's slice first, then 4 x 3 ViT blocks
    assert [b[2] for b in buckets] == [1, 3, 3, 3, 3], plan
    assert r0["nbytes"] == sum(b[1] for b in plan) and r0["nbytes"] > 4 * 30_000_000


def _sync_state_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sais_amd.parallel import GradSync
    sync = GradSync(world)
    torch.manual_seed(50 + rank)                       # every rank draws its own initial state, as loadModel does
    flat, proto = torch.rand(1000), torch.rand(1, 256)
    mine = flat.clone()
    sync.broadcast_initial_state([flat, proto])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(g, gathered[0]) for g in gathered) and (rank == 0) == torch.equal(flat, mine)
    tmax = sync.agree_max(7 + 4 * rank, "cpu")
    mean = sync.mean_scalar(1.0 + rank, "cpu")

    class M:
        _touched_T = 9
    sync._temporal = (M(), 5)                          # a 9-frame stream turned up, the exchange was sized for 5
    try:
        sync.flush_temporal()
        raised = False
    except RuntimeError:
        raised = True
    open(out + f".{rank}", "w").write(f"{same} {tmax} {mean} {raised}")
    dist.destroy_process_group()


def test_replicas_start_from_rank0_state_and_agree_on_scalars_gloo_world2(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "res")
    mp.spawn(_sync_state_worker, args=(2, 31500 + (os.getpid() % 2000), out), nprocs=2, join=True)
    for r in range(2):
        same, tmax, mean, raised = open(out + f".{r}").read().split()
        assert (same, int(tmax), float(mean), raised) == ("True", 11, 1.5, "True")


def _staged_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sais_amd.parallel import GradSync
    g = torch.Generator().manual_seed(7 + rank)
    flat = torch.randn(6_000_000, generator=g)                 # 24 MB "ViT" buffer: one bucket
    small1, small2, small3 = (torch.randn(n, generator=g) for n in (300, 500, 77))
    ref = [t.clone() for t in (flat, small1, small2, small3)]
    for r in ref:
        dist.all_reduce(r)
    # (a) flush() -> another small slice -> wait(): TWO packs in flight, both must be scattered back
    sync = GradSync(world)
    f, a, b, c = flat.clone(), small1.clone(), small2.clone(), small3.clone()
    sync._reduce(f); sync._reduce(a); sync._reduce(b)
    sync.flush()
    sync._reduce(c)                                            # a second backward call / TTA list element
    sync.wait()
    plan = [k for k, _, _ in sync.last_buckets]
    exact = all(torch.equal(x, r) for x, r in zip((f, a, b, c), ref))
    # (b) bf16 payload: half the bytes, sums within bf16 rounding of the fp32 exchange
    lo = GradSync(world, payload_dtype=torch.bfloat16)
    f2, a2 = flat.clone(), small1.clone()
    lo._reduce(f2); lo._reduce(a2)
    nbytes = lo.wait()
    rel = float((f2 - ref[0]).norm() / ref[0].norm()), float((a2 - ref[1]).norm() / ref[1].norm())
    open(out + f".{rank}", "w").write(f"{exact} {','.join(plan)} {nbytes} {rel[0]:.3e} {rel[1]:.3e} {f2.dtype == torch.float32}")
    dist.destroy_process_group()


def test_gradsync_two_packs_in_flight_and_bf16_payload_gloo_world2(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "res")
    mp.spawn(_staged_worker, args=(2, 33500 + (os.getpid() % 2000), out), nprocs=2, join=True)
    for r in range(2):
        exact, plan, nbytes, rel_f, rel_a, f32 = open(out + f".{r}").read().split()
        assert exact == "True" and plan == "bucket,packed,packed"
        assert int(nbytes) == 2 * (6_000_000 + 300) and f32 == "True"          # bf16 on the wire, fp32 in the buffers
        assert float(rel_f) <= 1e-2 and float(rel_a) <= 1e-2 and float(rel_f) > 0.0


class _StubModel:
    """Stands in for fullModel in the CPU test of the sharded window loop: deterministic host arithmetic on the batch."""
    modalities, importance_loss = "RGB-Flow", False

    def eval(self):
        return self

    def __call__(self, xs, fs, xlens, flens, task, xpads, fpads, domains):
        embs = [x.sum(dim=(1, 2))[:, :256] + 0.5 * f.sum(dim=(1, 2))[:, :256] for x, f in zip(xs, fs)]
        S = xs[0].shape[2] + 1
        attn = xs[0][:, 0, :1, :1].expand(-1, S, S) + torch.arange(S * S).view(1, S, S)
        return embs, attn


def _win_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sais_amd.inference import run_windows
    g = torch.Generator().manual_seed(5)
    rgb, flow = torch.randn(200, 384, generator=g), torch.randn(14, 384, generator=g)
    r, attn, imp = run_windows(_StubModel(), rgb, flow, videoname="v", batch_size=2, rank=rank, world_size=world)
    if rank == world - 1:                                  # every rank returns the merged lists, not only rank 0
        torch.save((r, attn, imp), out)
    dist.destroy_process_group()


def test_sharded_window_inference_merges_in_batch_order_gloo_world3(tmp_path):
    """window batches sharded over ranks, gathered in rank order == the single-process lists."""
    import torch.multiprocessing as mp
    from sais_amd.inference import run_windows
    from sais_amd.parallel import shard_range
    for n in (0, 1, 7, 8, 13):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    out = str(tmp_path / "win.pt")
    port = 29500 + ((os.getpid() + 7) % 2000)
    mp.spawn(_win_worker, args=(3, port, out), nprocs=3, join=True)
    r3, attn3, imp3 = torch.load(out, weights_only=False)
    g = torch.Generator().manual_seed(5)
    rgb, flow = torch.randn(200, 384, generator=g), torch.randn(14, 384, generator=g)
    r1, attn1, imp1 = run_windows(_StubModel(), rgb, flow, videoname="v", batch_size=2)
    assert len(r1["labels"]) == 13 and len(attn1) == 7            # (200 - 15) // 15 + 1 windows, 7 batches of <= 2
    assert r3["videonames"] == r1["videonames"] and len(r3["labels"]) == 13 and imp3 == imp1 == []
    for v in range(3):
        assert len(r3["reps"][v]) == 13 and all(torch.equal(a, b) for a, b in zip(r3["reps"][v], r1["reps"][v]))
    assert len(attn3) == 7 and all(torch.equal(a, b) for a, b in zip(attn3, attn1))


def test_window_sampler_and_tta_indices_follow_the_reference_quirks():
    from sais_amd.inference import flow_rows, gesture_windows, pad_collate_tta, sample_window, tta_indices
    wins = gesture_windows(512)
    assert len(wins) == 34 and wins[0] == (0, 15) and wins[-1] == (495, 510)          # (512-15)//15+1
    assert gesture_windows(14) == []
    i0, i1, i2 = tta_indices(0, 15)
    assert i0[0] == -1 and i0[-1] == 13 and [len(i0), len(i1), len(i2)] == [15, 12, 9]   # first window wraps (App. B.6)
    assert i1[0] == 2 and i2[0] == 5
    assert flow_rows(i0, 34) == [-1, 0] and flow_rows(tta_indices(15, 30)[0], 34) == [0, 1]
    assert flow_rows(tta_indices(495, 510)[0], 33) == [32]                               # row 33 filtered: >= len
    rgb, flow = torch.arange(512.).view(512, 1).repeat(1, 384), torch.arange(34.).view(34, 1).repeat(1, 384)
    xs, fs = sample_window(rgb, flow, 0, 15)
    assert xs[0][0, 0, 0].item() == 511.0 and xs[0][0, 1, 0].item() == 0.0              # index -1 -> last frame
    assert fs[0][0, :, 0].tolist() == [33.0, 0.0]
    c = pad_collate_tta([sample_window(rgb, flow, *wins[0]), sample_window(rgb, flow, *wins[1])])
    assert [tuple(t.shape) for t in c["x"]] == [(2, 1, 15, 384), (2, 1, 12, 384), (2, 1, 9, 384)]
    assert tuple(c["xpad"][0].shape) == (2, 1, 16) and not c["xpad"][0].any()
    assert c["flens"][0] == [2, 2] and tuple(c["f"][0].shape) == (2, 1, 2, 384)


def test_vectorised_window_collate_equals_the_per_window_path():
    """collate_windows_tta (one gather per TTA version and stream over ALL windows) builds exactly the tensors of
    pad_collate_tta([sample_window(...)]) — first-window wrap-around, ragged flow rows, rows filtered at the end of the
    flow features — for several video lengths and window subsets."""
    from sais_amd.inference import collate_windows_tta, gesture_windows, pad_collate_tta, sample_window
    g = torch.Generator().manual_seed(11)
    for n, nflow in ((512, 34), (512, 33), (77, 5), (45, 3), (15, 1), (200, 14)):
        rgb, flow = torch.randn(n, 384, generator=g), torch.randn(nflow, 384, generator=g)
        wins = gesture_windows(n)
        for sub in (wins, wins[:1], wins[1:4], wins[-2:]):
            if not sub:
                continue
            a = collate_windows_tta(rgb, flow, sub)
            b = pad_collate_tta([sample_window(rgb, flow, s, e) for s, e in sub])
            for key in ("x", "f", "xpad", "fpad"):
                assert len(a[key]) == 3 and all(x.shape == y.shape and torch.equal(x, y) for x, y in zip(a[key], b[key])), (n, key)
            assert a["xlens"] == b["xlens"] and a["flens"] == b["flens"]


def test_stacked_window_collate_equals_the_per_version_tensors():
    """collate_windows_merged (ONE stacked batch for the single encoder pass of the inference path) holds, sequence by sequence, the
    rows, zero padding and key masks of pad_collate_tta([sample_window(...)]) — wrap-around of the first window, one or two flow
    rows, flow rows filtered at the end of the flow features."""
    from sais_amd.inference import collate_windows_merged, gesture_windows, pad_collate_tta, sample_window
    g = torch.Generator().manual_seed(12)
    for n, nflow in ((512, 34), (512, 33), (77, 5), (45, 3), (15, 1), (200, 14), (200, 12)):
        rgb, flow = torch.randn(n, 384, generator=g), torch.randn(nflow, 384, generator=g)
        wins = gesture_windows(n)
        for sub in (wins, wins[:1], wins[1:4], wins[-2:]):
            if not sub:
                continue
            X, P, where = collate_windows_merged(rgb, flow, sub)
            b = pad_collate_tta([sample_window(rgb, flow, s, e) for s, e in sub])
            B = len(sub)
            assert X.shape == (6 * B, 1, 15, 384) and P.shape == (6 * B, 16) and P.dtype == torch.uint8
            for key in ("x", "f"):
                for v in range(3):
                    o, k, T = where[(key, v)]
                    ref, mask, lens = b[key][v], b[key + "pad"][v], b[key + "lens"][v]
                    Tv = ref.shape[2]
                    assert k == B and T >= Tv
                    assert torch.equal(X[o:o + B, 0, :Tv], ref[:, 0]) and not X[o:o + B, 0, Tv:].any(), (n, nflow, key, v)
                    assert torch.equal(P[o:o + B, :Tv + 1].bool(), mask[:, 0]) and P[o:o + B, Tv + 1:].all()
                    assert [int(t) for t in (1 - P[o:o + B, 1:].long()).sum(1)] == lens


def test_window_sampler_matches_the_reference_dataset(golden):
    """Frame / flow-row indices of every Custom_inference window and TTA version vs tests/golden/sampler.npz, which
    was produced by the reference's own VideoDataset.__getitem__ (make_golden_sampler.py)."""
    import numpy as np
    from sais_amd.inference import gesture_windows, sample_window
    g = golden("sampler")
    for name, n in (("n512", 512), ("n77", 77), ("n45", 45), ("n15", 15)):
        nflow = max(n // 15, 1)
        rgb = torch.arange(float(n)).view(n, 1).repeat(1, 384)
        flow = torch.arange(float(nflow)).view(nflow, 1).repeat(1, 384)
        wins = gesture_windows(n)
        assert len(wins) == int(g[name + "/nwindows"])
        for w, (s, e) in enumerate(wins):
            xs, fs = sample_window(rgb, flow, s, e)
            for v in range(3):
                assert np.array_equal(xs[v][0, :, 0].numpy().astype(np.int64), g[f"{name}/w{w}/rgb{v}"]), (name, w, v)
                assert np.array_equal(fs[v][0, :, 0].numpy().astype(np.int64), g[f"{name}/w{w}/flow{v}"]), (name, w, v)
            assert xs[0].shape[1] == int(g[f"{name}/w{w}/imp_len"])


def test_counter_files_go_stale_when_a_kernel_source_changes(tmp_path):
    """bench.py's `roofline.traffic` comes from profiles/pmc_traffic.json, collected offline.  The file carries the hash of the kernel
    sources it was collected on; a one-comment edit of a copy of gemm.hip must turn the field off (`traffic_stale`)."""
    import json
    import shutil
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for sub in (("sais_amd", "csrc"), ("include",)):
        os.makedirs(tmp_path.joinpath(*sub))
    for f in os.listdir(os.path.join(root, "sais_amd", "csrc")):
        if f.endswith((".hip", ".hpp")) or f == "Makefile":
            shutil.copy(os.path.join(root, "sais_amd", "csrc", f), tmp_path / "sais_amd" / "csrc" / f)
    shutil.copy(os.path.join(root, "include", "sais_hip.h"), tmp_path / "include" / "sais_hip.h")
    os.makedirs(tmp_path / "profiles")
    h = bench.kernel_source_hash(str(tmp_path))
    assert h == bench.kernel_source_hash(root)                       # same bytes, same hash
    json.dump({"gemm_tn_grouped": {"hbm_bytes_per_launch": 1}, "_provenance": {"kernel_source_hash": h}},
              open(tmp_path / "profiles" / "pmc_traffic.json", "w"))
    d, stale = bench.load_pmc("pmc_traffic.json", str(tmp_path))
    assert d["gemm_tn_grouped"]["hbm_bytes_per_launch"] == 1 and stale is False
    with open(tmp_path / "sais_amd" / "csrc" / "gemm.hip", "a") as fh:
        fh.write("// a comment\n")
    d, stale = bench.load_pmc("pmc_traffic.json", str(tmp_path))
    assert stale is True
    json.dump({"gemm_tn_grouped": {"hbm_bytes_per_launch": 1}, "_provenance": {"head": "205f2ce"}},       # pre-round-6 file: no hash
              open(tmp_path / "profiles" / "pmc_traffic.json", "w"))
    assert bench.load_pmc("pmc_traffic.json", str(tmp_path))[1] is True


def test_feature_extractor_shapes_per_remainder():
    """FeatureExtractor.shape_for: the padded shape of a forward pass for `left` remaining frames under the three tail policies
    (no GPU: only the host arithmetic; the replay / eager bit-identity is tests/test_inference_gpu.py)."""
    from sais_amd.inference import FeatureExtractor

    class _V(torch.nn.Module):
        pass
    one = FeatureExtractor(_V(), batch_size=256)
    fixed = FeatureExtractor(_V(), batch_size=256, tail_batch=64)
    fit = FeatureExtractor(_V(), batch_size=256, tail_batch="fit")
    fit8 = FeatureExtractor(_V(), batch_size=256, tail_batch="fit", tail_round=8)
    for left, want in ((300, (256, 256, 256, 256)), (256, (256, 256, 256, 256)), (255, (256, 64, 256, 256)),
                       (100, (256, 64, 100, 104)), (34, (256, 64, 34, 40)), (33, (256, 64, 34, 40)), (1, (256, 64, 2, 8))):
        assert (one.shape_for(left), fixed.shape_for(left), fit.shape_for(left), fit8.shape_for(left)) == want, left
