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
    ap.add_argument('--checkpoint', default=None, type=str, help='dino_deitsmall16_pretrain.pth (default: dino-main/outputs/)')
    args = ap.parse_args()
    if args.arch != 'vit_small' or args.patch_size != 16:
        raise SystemExit('the MI355X kernels implement vit_small / patch 16 only')
    if args.optical_flow or args.segmentation or args.segmentation_to_reps:
        raise SystemExit('RAFT optical flow / segmentation are out of scope of this build (SURVEY.md §2)')
    t0 = time.time()
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
    fx = FeatureExtractor(vit, batch_size=min(args.batch_size_per_gpu, 256), use_graph=True)
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
