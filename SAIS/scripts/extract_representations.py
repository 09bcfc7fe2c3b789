#!/usr/bin/env python3
"""Drop-in for the hot-path branch of the reference's SAIS/scripts/extract_representations.py (:351-378, flags
:410-435): DINO ViT-S/16 CLS features of every frame of a video -> results/<model_type>_[Flow]RepsAndLabels.{h5|npz}.
The ViT runs on the MI355X kernels (sais_amd.vit) inside a hipGraph.  Flags keep the reference's names; the
`--arch` choices no longer call torch.hub (a network call at parser build time, :416)."""
import argparse
import glob
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from SAIS.scripts._features_io import save_reps  # noqa: E402

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)          # :148


FLOW_MARKER = '.flows_complete.json'                               # written by --optical_flow when a video's flows are all saved


class FrameError(Exception):
    """A frame this rank's shard cannot use.  Raised locally, reported collectively (every rank leaves together)."""


def frame_batches(folder, dev, chunk=256, rank=0, world=1):
    """SurgDataset.__getitem__ (dino-main/main_dino.py:295-316) + the transform of :158-162, with the arithmetic on the
    GPU: JPEGs are decoded on the host (PIL), pushed as uint8 and turned into the float32 [n,3,224,224] ViT input by
    sais_amd.preprocess (CenterCrop(0.8 H, 0.8 W) -> Resize((224,224)) -> ToTensor -> Normalize, bit-identical to the
    torchvision 0.9.0 / Pillow pipeline of the reference).  Yields device tensors of up to `chunk` frames.
    world > 1: rank r decodes and embeds a contiguous range of the (sorted) frame files only."""
    from sais_amd.parallel import shard_range
    from PIL import Image
    from sais_amd.preprocess import FramePreprocessor
    plans, buf, geom = {}, [], None

    def flush():
        h, w = geom
        if geom not in plans:
            plans[geom] = FramePreprocessor(h, w, 0.8, 0.8, MEAN, STD, device=dev)
        return plans[geom](torch.from_numpy(np.stack(buf)))

    files = sorted(glob.glob(os.path.join(folder, '*.jpg')))
    lo, hi = shard_range(len(files), rank, world)
    for p in files[lo:hi]:
        with Image.open(p) as img:
            if img.mode != 'RGB':                    # the reference drops the result of img.convert('RGB') (:297)
                raise FrameError(f'{p}: mode {img.mode}; the pipeline expects RGB frames')
            a = np.asarray(img)
        if buf and (a.shape[:2] != geom or len(buf) == chunk):
            yield flush()
            buf = []
        geom = a.shape[:2]
        buf.append(a)
    if buf:
        yield flush()


def extract_flows(args, t0):
    """--optical_flow (reference :30-143,221-288 and the __main__ branch :472-497): for every row of
    paths/<dataset>_FlowPaths.csv (frames 15 apart, written by generate_paths.py) estimate the flow with RAFT, colour-code it
    (flow_to_rgb) and save flows/<label>/flows_<nflow:08d>.jpg with nflow = frame number // jump_size.  RAFT runs on the GPU
    (sais_amd.raft: correlation volume on the HIP kernels); PARITY UNPINNED — ptlflow 0.2.5 / the 'things' checkpoint are
    absent.  Without --raft_checkpoint the stage REFUSES to generate anything (exit 2) unless --raft_random_weights asks
    for seeded random weights explicitly (smoke runs): flow maps of an untrained RAFT are noise, and they would land in the
    folder the flow stream reads.

    A video is skipped, as in the reference (:487), when its flows folder is COMPLETE: it either carries this stage's marker
    (`.flows_complete.json`: count + which weights) with as many flows_*.jpg as the CSV asks for, or — no marker — it holds
    exactly that many files (flows the user supplied).  A folder without marker and with fewer files (an interrupted run)
    is regenerated, and so is a folder whose marker says "random" once a real checkpoint is given."""
    import csv
    import json
    from PIL import Image
    dev = torch.device('cuda', args.local_rank)
    weights = ('checkpoint:' + os.path.basename(args.raft_checkpoint)) if args.raft_checkpoint else 'random-seed-0'
    model = None

    def get_model():
        from sais_amd.raft import RAFT
        torch.manual_seed(0)
        m = RAFT(iters=args.raft_iters)
        if args.raft_checkpoint:
            sd = torch.load(args.raft_checkpoint, map_location='cpu')
            m.load_state_dict({k.replace('module.', '', 1): v for k, v in sd.items()}, strict=True)
        else:
            print('[flow] --raft_random_weights: seeded random RAFT weights, the flow maps are NOT optical flow '
                  '(parity unpinned; see sais_amd/raft.py)')
        return m.to(dev).eval()

    def complete(label, want):
        folder = os.path.join(args.data_path, 'flows', label)
        if not os.path.isdir(folder):
            return False
        have = len(glob.glob(os.path.join(folder, 'flows_*.jpg')))
        marker = os.path.join(folder, FLOW_MARKER)
        if not os.path.exists(marker):
            return have >= want                      # user-supplied flows; a partial folder of an interrupted run is not
        with open(marker) as fh:
            info = json.load(fh)
        if info.get('weights', '').startswith('random') and args.raft_checkpoint:
            return False                             # junk of a smoke run: regenerate with the real weights
        return have >= want and info.get('count') == want

    from sais_amd.raft import flow_image_uint8, flow_to_rgb
    for dataset in args.data_list:
        jump = 30 if dataset in ('VUA_Lab', 'DVC_UCL') else 15                       # :488-493
        with open(os.path.join(args.data_path, 'paths', '%s_FlowPaths.csv' % dataset)) as fh:
            rows = list(csv.DictReader(fh))
        if args.video:
            rows = [r for r in rows if r['label'] == args.video]
        # files per video = DISTINCT flow numbers (two CSV rows that map to the same frame // jump overwrite one file: counting rows
        # would make such a folder look incomplete on every later run)
        def nflow_of(r):
            p1 = r['path1'].replace('\\', '/')
            return int(p1.split('frames_' if 'frames' in p1 else 'frame_')[-1].strip('.jpg')) // jump
        distinct = {}
        for r in rows:
            distinct.setdefault(r['label'], set()).add(nflow_of(r))
        want = {lab: len(v) for lab, v in distinct.items()}
        done = {lab for lab, n in want.items() if complete(lab, n)}
        rows = [r for r in rows if r['label'] not in done]
        if rows and not args.raft_checkpoint and not args.raft_random_weights:
            raise SystemExit('[flow] %d flow maps of %s are missing and no --raft_checkpoint was given.  Pass the RAFT "things" '
                             'state dict (main.sh: RAFT_CHECKPOINT=...), put your own flows_*.jpg under flows/<video>/, or ask '
                             'for seeded random weights explicitly with --raft_random_weights (smoke runs only).'
                             % (len(rows), sorted(set(want) - done)))
        if rows and model is None:
            model = get_model()
        for lab in set(want) - done:                 # a stale marker / partial folder must not survive a failed run
            mk = os.path.join(args.data_path, 'flows', lab, FLOW_MARKER)
            had_marker = os.path.exists(mk)
            if had_marker:
                os.remove(mk)
            mine = glob.glob(os.path.join(args.data_path, 'flows', lab, 'flows_*.jpg'))
            if mine and not had_marker and args.raft_random_weights and not args.raft_checkpoint:
                # a marker-less partial folder: the leftovers of an interrupted run, or files the user put there.  They are
                # regenerated (an interrupted run must not count as done) — but say so loudly when the replacement is noise
                print('[flow] WARNING: flows/%s holds %d flows_*.jpg without %s and fewer than the %d the video needs: they are '
                      'OVERWRITTEN with maps from seeded RANDOM weights (smoke runs only; pass --raft_checkpoint for real flow)'
                      % (lab, len(mine), FLOW_MARKER, want[lab]), file=sys.stderr)
        bs = max(1, args.batch_size_per_gpu)
        nsaved = 0
        saved = {}                                   # label -> flow numbers written in this run
        for i in range(0, len(rows), bs):
            chunk = rows[i:i + bs]
            pairs, nflows = [], []
            for r in chunk:
                p1, p2 = (r[k].replace('\\', '/') for k in ('path1', 'path2'))
                name = 'frames_' if 'frames' in p1 else 'frame_'
                nflows.append(int(p1.split(name)[-1].strip('.jpg')) // jump)          # :104
                if args.synthetic_frames:
                    g = torch.Generator().manual_seed(nflows[-1])
                    base = torch.rand(3, 232, 232, generator=g)
                    pairs.append((base[:, :224, :224], base[:, 4:228, 2:226]))
                else:
                    fr = []
                    for p in (p1, p2):
                        with Image.open(os.path.join(args.data_path, p)) as im:
                            a = torch.from_numpy(np.asarray(im.convert('RGB'))).permute(2, 0, 1).float() / 255.0
                        fr.append(a.flip(0))                                          # cv.imread order (BGR), as the reference feeds it
                    pairs.append(tuple(fr))
            shapes = {tuple(a.shape) for a, _ in pairs}
            groups = [[k for k, (a, _) in enumerate(pairs) if tuple(a.shape) == sh] for sh in shapes]
            for idx in groups:                                                         # frames of one size per batch
                i1 = torch.stack([pairs[k][0] for k in idx]).to(dev)
                i2 = torch.stack([pairs[k][1] for k in idx]).to(dev)
                flows = model(i1, i2)
                for k, fl in zip(idx, flows):
                    img = Image.fromarray(flow_image_uint8(flow_to_rgb(fl)))          # :245-249
                    out = os.path.join(args.data_path, 'flows', chunk[k]['label'])
                    os.makedirs(out, exist_ok=True)
                    img.save(os.path.join(out, 'flows_%08d.jpg' % nflows[k]))         # :254-262
                    saved.setdefault(chunk[k]['label'], set()).add(nflows[k])
                    nsaved += 1
        for lab in set(want) - done:                 # complete only when every distinct flow number of the video was written
            if len(saved.get(lab, ())) != want[lab]:
                print(f'[flow] {lab}: {len(saved.get(lab, ()))} of {want[lab]} flow maps written: folder left unmarked (incomplete)')
                continue
            with open(os.path.join(args.data_path, 'flows', lab, FLOW_MARKER), 'w') as fh:
                json.dump({'count': want[lab], 'weights': weights, 'iters': args.raft_iters}, fh)
        print(f'[flow] {dataset}: {nsaved} flow maps saved' + (f' ({len(done)} videos already had flows)' if done else ''))
    print('All Flows Saved!')
    print('Time taken (s): %.3f' % (time.time() - t0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arch', default='vit_small', type=str, choices=['vit_tiny', 'vit_small', 'vit_base'])
    ap.add_argument('--patch_size', default=16, type=int)
    ap.add_argument('--drop_path_rate', type=float, default=0.1)
    ap.add_argument('--model_type', default='ViT_SelfSupervised_ImageNet', type=str)
    ap.add_argument('--batch_size_per_gpu', default=64, type=int)
    ap.add_argument('--data_path', default='./SAIS/', type=str)
    ap.add_argument('--data_list', default=['Custom'], nargs='+')
    ap.add_argument('--save_type', default='h5', choices=['dict', 'h5'])
    ap.add_argument('--optical_flow', action='store_true')
    ap.add_argument('--segmentation', action='store_true')
    ap.add_argument('--optical_flow_to_reps', action='store_true')
    ap.add_argument('--segmentation_to_reps', action='store_true')
    ap.add_argument('--local_rank', '--local-rank', default=0, type=int)
    ap.add_argument('--video', default=None, type=str, help='video label (folder under images/ or flows/)')
    ap.add_argument('--synthetic_frames', default=0, type=int, help='use N seeded synthetic frames instead of JPEGs')
    ap.add_argument('--tail_batch', default='fit', type=str,
                    help="shape of the last, partial batch of a video: 'fit' (default) = its own size (rounded up to an even "
                         "number of frames; fastest), N = padded to a fixed second shape of N frames, 0 = padded to "
                         "--batch_size_per_gpu.  With 0 / N every frame runs the same kernels whatever the number of ranks, so the "
                         "feature files are bit-identical across world sizes; with 'fit' they agree to bf16 rounding")
    ap.add_argument('--checkpoint', default=None, type=str, help='dino_deitsmall16_pretrain.pth (default: dino-main/outputs/)')
    ap.add_argument('--raft_checkpoint', default=None, type=str,
                    help='--optical_flow: a RAFT state dict with the published key names (e.g. raft-things.pth); default: '
                         'seeded random weights (ptlflow and its checkpoint are unreachable offline: parity unpinned)')
    ap.add_argument('--raft_random_weights', action='store_true',
                    help='--optical_flow without a checkpoint: generate with seeded random weights anyway (smoke runs; the '
                         'folder is marked so that a later run with a checkpoint regenerates it)')
    ap.add_argument('--raft_iters', default=12, type=int)
    args = ap.parse_args()
    if args.arch != 'vit_small' or args.patch_size != 16:
        raise SystemExit('the MI355X kernels implement vit_small / patch 16 only')
    if args.segmentation or args.segmentation_to_reps:
        raise SystemExit('segmentation is out of scope of this build (SURVEY.md §2)')
    t0 = time.time()
    if args.optical_flow:
        return extract_flows(args, t0)
    from sais_amd.inference import FeatureExtractor
    from sais_amd.model_io import load_vit
    from sais_amd.parallel import gather_in_rank_order, init_from_env, shard_range
    rank, world, local = init_from_env()                 # torch.distributed.run: frames of a video are sharded over the ranks
    dev = torch.device('cuda', local if world > 1 else args.local_rank)
    ckpt = args.checkpoint or os.path.join(args.data_path, 'scripts', 'dino-main', 'outputs', 'dino_deitsmall16_pretrain.pth')
    if not os.path.exists(ckpt):
        print(f'[extract] {ckpt} not found: using seeded random ViT-S/16 weights (no network in this environment)')
        ckpt = None
    torch.manual_seed(0)
    vit = load_vit(ckpt, device=dev, drop_path_rate=args.drop_path_rate)
    flow = args.optical_flow_to_reps
    sub = 'flows' if flow else 'images'
    videos = [args.video] if args.video else sorted(os.listdir(os.path.join(args.data_path, sub)))
    fx = FeatureExtractor(vit, batch_size=min(args.batch_size_per_gpu, 256), use_graph=True,
                          tail_batch='fit' if args.tail_batch == 'fit' else (int(args.tail_batch) or None))
    reps = {}
    for v in videos:
        err = None
        if args.synthetic_frames:
            g = torch.Generator().manual_seed(1 if flow else 0)
            n = max(1, args.synthetic_frames // 15) if flow else args.synthetic_frames     # flow maps: every 15th frame
            u8 = torch.randint(0, 256, (n, 3, 224, 224), generator=g, dtype=torch.uint8).float() / 255.0
            frames = (u8 - torch.tensor(MEAN).view(1, 3, 1, 1)) / torch.tensor(STD).view(1, 3, 1, 1)
            lo, hi = shard_range(n, rank, world)
            mine = fx(frames[lo:hi].to(dev)).cpu() if hi > lo else torch.empty(0, 384)
        else:
            try:
                parts = [fx(b).cpu() for b in frame_batches(os.path.join(args.data_path, sub, v), dev, rank=rank, world=world)]
                mine = torch.cat(parts) if parts else torch.empty(0, 384)
            except Exception as e:           # ANY rank-local failure (bad frame mode, PIL decode error, OSError, a geometry
                # error inside fx): the other ranks are about to enter the gather below, so it is reported there
                err = str(e) if isinstance(e, FrameError) else f'rank {rank}: {type(e).__name__}: {e}'
                mine = torch.empty(0, 384)
        # the error flag travels WITH the payload, so that one rank's bad frame ends the job on every rank at once instead of
        # leaving the others in all_gather_object until the process-group timeout
        gathered = gather_in_rank_order((err, mine), world)
        errs = [e for e, _ in gathered if e]
        if errs:
            raise SystemExit(errs[0])
        full = torch.cat([m for _, m in gathered])                    # rank order = frame order
        if full.shape[0] == 0:
            raise SystemExit(f'no frames under {os.path.join(args.data_path, sub, v)}')
        reps[v] = full.numpy()
        if rank == 0:
            print(f'[extract] {v}: {reps[v].shape[0]} frames -> {reps[v].shape}' + (f' ({world} ranks)' if world > 1 else ''))
    name = '%s_%sRepsAndLabels' % (args.model_type, 'Flow' if flow else '')
    if rank == 0:                                                     # rank 0 writes (train.py:98)
        print('[extract] saved', save_reps(args.data_path, name, reps))
        print('Time taken (s): %.3f' % (time.time() - t0))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
