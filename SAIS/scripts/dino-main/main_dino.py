#!/usr/bin/env python3
"""DINO pre-training of the ViT-S/16 frame encoder on MI355X — same command line, log and checkpoint files as the
reference's SAIS/scripts/dino-main/main_dino.py (flags: tests/golden/dino_args.json is the reference parser's own
dump), driving sais_amd.dino (hand-written gfx950 kernels, one process per GPU, RCCL for the gradient and centre
all-reduces).

    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 SAIS/scripts/dino-main/main_dino.py \
        --data_path <root with paths/<dataset>_Paths.csv> --output_dir <dir> [--epochs ...]

Kept: every flag, the linear lr scaling rule lr * global_batch / 256, the cosine schedules, `checkpoint.pth` /
`checkpoint%04d.pth` / `log.txt`, resume from `--output_dir/checkpoint.pth`.  Differences: `--use_fp16` is accepted and
ignored (bf16 MFMA operands with fp32 master weights need no loss scaler); `--arch` must be vit_small (deit_small),
`--optimizer` adamw, `--use_bn_in_head` False; the dataset list the reference hard-codes (:353) is the default of the
extra `--datasets` flag, its hard-coded frame root './SAIS' (:288) the default of `--frames_root`.
"""
import argparse
import datetime
import json
import math
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from sais_amd import dino  # noqa: E402
from sais_amd.dino_data import DataAugmentationDINO, SurgDataset  # noqa: E402


def bool_flag(s):
    """utils.bool_flag (utils.py:201-212)."""
    if s.lower() in {"off", "false", "0"}:
        return False
    if s.lower() in {"on", "true", "1"}:
        return True
    raise argparse.ArgumentTypeError("invalid value for a boolean flag")


FLAGS = [  # (flag, type, default)   — the reference parser, main_dino.py:48-141
    ("--arch", str, "vit_small"), ("--patch_size", int, 16), ("--out_dim", int, 65536),
    ("--norm_last_layer", bool_flag, True), ("--momentum_teacher", float, 0.996), ("--use_bn_in_head", bool_flag, False),
    ("--warmup_teacher_temp", float, 0.04), ("--teacher_temp", float, 0.04), ("--warmup_teacher_temp_epochs", int, 0),
    ("--use_fp16", bool_flag, True), ("--weight_decay", float, 0.04), ("--weight_decay_end", float, 0.4),
    ("--clip_grad", float, 3.0), ("--batch_size_per_gpu", int, 64), ("--epochs", int, 100),
    ("--freeze_last_layer", int, 1), ("--lr", float, 0.0005), ("--warmup_epochs", int, 10), ("--min_lr", float, 1e-6),
    ("--optimizer", str, "adamw"), ("--drop_path_rate", float, 0.1), ("--local_crops_number", int, 8),
    ("--data_path", str, "/path/to/imagenet/train/"), ("--output_dir", str, "."), ("--saveckp_freq", int, 20),
    ("--seed", int, 0), ("--num_workers", int, 10), ("--dist_url", str, "env://"), ("--local_rank", int, 0),
]


def get_args_parser():
    p = argparse.ArgumentParser("DINO", add_help=False)
    for flag, typ, default in FLAGS:
        p.add_argument(flag, type=typ, default=default)
    p.add_argument("--global_crops_scale", type=float, nargs="+", default=(0.4, 1.))
    p.add_argument("--local_crops_scale", type=float, nargs="+", default=(0.05, 0.4))
    for flag in ("--optical_flow_to_reps", "--segmentation_to_reps", "--optical_flow", "--segmentation"):
        p.add_argument(flag, default=False, action="store_true")
    p.add_argument("--task", default="DINO")
    # not in the reference: what it hard-codes
    p.add_argument("--datasets", nargs="+", default=["VUA", "VUA_Gronau", "VUA_HMH"])
    p.add_argument("--frames_root", default="./SAIS")
    return p


def init_distributed(args):
    """utils.init_distributed_mode (utils.py:468-500): torchrun / launch env, else a single process."""
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        args.rank, args.world_size = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        args.gpu = int(os.environ.get("LOCAL_RANK", args.local_rank))
    else:
        args.rank, args.gpu, args.world_size = 0, 0, 1
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
    torch.cuda.set_device(args.gpu)
    dist.init_process_group("nccl", init_method=args.dist_url, world_size=args.world_size, rank=args.rank,
                            device_id=torch.device("cuda", args.gpu))
    dist.barrier()


def collate(batch):
    crops = [torch.stack([b[0][i] for b in batch]) for i in range(len(batch[0][0]))]
    return crops, [b[1] for b in batch], [b[2] for b in batch]


def train_dino(args):
    init_distributed(args)
    torch.manual_seed(args.seed)
    main = args.rank == 0
    if args.arch.replace("deit", "vit") != "vit_small" or args.patch_size != 16 or args.optimizer != "adamw" or args.use_bn_in_head:
        raise NotImplementedError("MI355X path: --arch vit_small --patch_size 16 --optimizer adamw --use_bn_in_head false")
    dev = torch.device("cuda", args.gpu)
    transform = DataAugmentationDINO(args.global_crops_scale, args.local_crops_scale, args.local_crops_number,
                                     seed=args.seed * 1000 + args.rank)
    dataset = SurgDataset(args.data_path, args.datasets, transform, frames_root=args.frames_root)
    sampler = torch.utils.data.DistributedSampler(dataset, shuffle=True)
    loader = torch.utils.data.DataLoader(dataset, sampler=sampler, batch_size=args.batch_size_per_gpu,
                                         num_workers=args.num_workers, pin_memory=True, drop_last=True, collate_fn=collate)
    if main:
        print(f"Data loaded: there are {len(dataset)} images.")
    student, teacher = dino.build_student_teacher(args.out_dim, args.drop_path_rate, args.norm_last_layer, dev)
    for m in (student, teacher):                                     # DDP's start-up broadcast: every rank = rank 0
        for t in list(m.parameters()) + list(m.buffers()):
            dist.broadcast(t.data, 0)
    dino_loss = dino.DINOLoss(args.out_dim, args.local_crops_number + 2, args.warmup_teacher_temp, args.teacher_temp,
                              args.warmup_teacher_temp_epochs, args.epochs).to(dev)
    optimizer = dino.DINOOptimizer(student, teacher)
    niter = len(loader)
    lr_schedule = dino.cosine_scheduler(args.lr * (args.batch_size_per_gpu * args.world_size) / 256., args.min_lr,
                                        args.epochs, niter, warmup_epochs=args.warmup_epochs)
    wd_schedule = dino.cosine_scheduler(args.weight_decay, args.weight_decay_end, args.epochs, niter)
    momentum_schedule = dino.cosine_scheduler(args.momentum_teacher, 1, args.epochs, niter)
    start_epoch = 0
    ckpt_path = os.path.join(args.output_dir, "checkpoint.pth")
    if os.path.isfile(ckpt_path):                                    # utils.restart_from_checkpoint
        ck = torch.load(ckpt_path, map_location="cpu", weights_only=False)
        with torch.no_grad():                                        # flat buffers exist after one forward
            probe = [torch.zeros(1, 3, 224, 224, device=dev)] * 2
            student(probe), teacher(probe)
        start_epoch = dino.load_checkpoint(ck, student, teacher, optimizer, dino_loss)
        if main:
            print(f"=> resumed from {ckpt_path} at epoch {start_epoch}")
    t_start = time.time()
    if main:
        print("Starting DINO training !")
    for epoch in range(start_epoch, args.epochs):
        sampler.set_epoch(epoch)
        total, count, t0 = torch.zeros((), device=dev), 0, time.time()
        finite = torch.ones((), device=dev)                          # device-side AND of isfinite(loss), every iteration
        for i, (images, _, _) in enumerate(loader):
            it = niter * epoch + i
            images = [im.to(dev, non_blocking=True) for im in images]
            loss, _ = dino.train_step(student, teacher, dino_loss, optimizer, images, it, epoch, lr_schedule, wd_schedule,
                                      momentum_schedule, clip_grad=args.clip_grad, freeze_last_layer=args.freeze_last_layer)
            total += loss
            finite *= torch.isfinite(loss).to(finite.dtype)
            count += 1
            if i % 10 == 0:                                          # the reference logs every 10 iterations
                lv = loss.item()
                if not math.isfinite(lv):
                    print("Loss is {}, stopping training".format(lv))
                    sys.exit(1)
                if main:
                    print(f"Epoch: [{epoch}/{args.epochs}]  [{i}/{niter}]  loss: {lv:.6f}  lr: {lr_schedule[it]:.6f}  "
                          f"wd: {wd_schedule[it]:.6f}  {(time.time() - t0) / (i + 1):.4f} s / it")
        dist.all_reduce(total)
        dist.all_reduce(finite, op=dist.ReduceOp.MIN)
        if finite.item() == 0:                                       # the reference checks every iteration (main_dino.py:540-542);
            print("Loss went non-finite in epoch {}, stopping training".format(epoch))   # here: before checkpoint.pth is
            sys.exit(1)                                              # overwritten with poisoned weights
        stats = {"loss": (total / max(count * args.world_size, 1)).item(), "lr": float(lr_schedule[min(it, len(lr_schedule) - 1)]),
                 "wd": float(wd_schedule[min(it, len(wd_schedule) - 1)])} if count else {}
        if main:
            save = dino.checkpoint_dict(student, teacher, optimizer, dino_loss, epoch + 1, args)
            torch.save(save, ckpt_path)
            if args.saveckp_freq and epoch % args.saveckp_freq == 0:
                torch.save(save, os.path.join(args.output_dir, f"checkpoint{epoch:04}.pth"))
            with open(os.path.join(args.output_dir, "log.txt"), "a") as f:
                f.write(json.dumps({**{f"train_{k}": v for k, v in stats.items()}, "epoch": epoch}) + "\n")
        dist.barrier()
    if main:
        print("Training time {}".format(str(datetime.timedelta(seconds=int(time.time() - t_start)))))
    dist.destroy_process_group()


if __name__ == "__main__":
    parser = argparse.ArgumentParser("DINO", parents=[get_args_parser()])
    args = parser.parse_args()
    os.makedirs(args.output_dir, exist_ok=True)
    train_dino(args)
