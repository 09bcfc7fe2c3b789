#!/usr/bin/env python3
"""Drop-in for the reference's SAIS/scripts/run_experiments.py (:19-121) on MI355X: same flags, same
params/Fold_<k>/ layout.
  --inference (what main.sh:27 runs): (params.zip, prototypes.zip) in, reps_and_labels_<ph> / attention_<ph> /
      importance_<ph> out (train.py:113-119), windows from paths/Custom_Paths.csv.
  without --inference: sais_amd.train.trainModel (train.py:18-121) on the feature files under results/ and the
      annotated windows of paths/<dataset>_Annotations.csv; writes params / prototypes / metrics / reps_and_labels.
Only -m ViT -t Prototypes -dt reps is on this build's path."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from SAIS.scripts._features_io import load_reps  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument('-p', '--path', type=str)
    p.add_argument('-data', '--dataset_name', type=str)
    p.add_argument('-d', '--domain_name', type=str)
    p.add_argument('-m', '--model', type=str, default='R3D')
    p.add_argument('-enc', '--encoder_params', type=str, default='ViT_SelfSupervised_ImageNet')
    p.add_argument('-dim', '--rep_dim', type=int, default=512)
    p.add_argument('-mod', '--modalities', type=str)
    p.add_argument('-bs', '--batch_size', type=int, default=1)
    p.add_argument('-lr', '--learning_rate', type=float)
    p.add_argument('-tf', '--training_fraction', type=float, default=1)
    p.add_argument('-fe', '--freeze_encoder', default=False, action='store_true')
    p.add_argument('-t', '--task', type=str)
    p.add_argument('-nc', '--nclasses', type=int)
    p.add_argument('-bc', '--balance_classes', default=False, action='store_true')
    p.add_argument('-bg', '--balance_groups', default=False, action='store_true')
    p.add_argument('-sg', '--single_group', default=False, action='store_true')
    p.add_argument('-sa', '--self_attention', default=False, action='store_true')
    p.add_argument('-il', '--importance_loss', default=False, action='store_true')
    p.add_argument('-domains', '--domains', nargs='+', type=str)
    p.add_argument('-ph', '--phases', nargs='+')
    p.add_argument('-dt', '--data_type', type=str)
    p.add_argument('-e', '--nepochs', type=int)
    p.add_argument('-f', '--nfolds', type=int)
    p.add_argument('-i', '--inference', default=False, action='store_true')
    p.add_argument('--local_rank', '--local-rank', type=int, default=0)
    a = p.parse_args()
    print('Modalities: %s' % a.modalities)
    print('Self Attention: %s' % str(a.self_attention))
    t0 = time.time()
    if a.model != 'ViT' or a.task != 'Prototypes' or a.data_type != 'reps':
        raise SystemExit('this build covers -m ViT -t Prototypes -dt reps only (SURVEY.md §8)')
    if not a.inference:
        from sais_amd.parallel import init_from_env
        from sais_amd.train import trainModel
        # the reference pins world_size = 1 here (run_experiments.py:112, its DDP is commented out); under
        # torch.distributed.run this CLI trains data-parallel: every rank a shard of the training windows, gradients
        # exchanged by sais_amd.parallel.GradSync, rank 0 writes the files
        rank, world, local = init_from_env()
        for domain in a.domains:
            for fold in range(a.nfolds):
                savepath = os.path.join(a.path, 'params/Fold_%i' % fold)             # getSavepath :82-83
                print('***** \n Savepath: %s \n *****' % savepath)
                trainModel(local if world > 1 else a.local_rank, world, a.path, savepath, a.dataset_name, a.data_type,
                           a.batch_size, a.nclasses,
                           domain, a.phases, a.learning_rate, a.modalities, a.freeze_encoder, False, a.task,
                           a.balance_classes, a.balance_groups, a.single_group, 'None', a.self_attention,
                           a.importance_loss, a.model, a.encoder_params, 5, 1, 0, a.rep_dim, a.nepochs, fold,
                           a.training_fraction)
        if world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        print('Time taken (s): %.3f' % (time.time() - t0))
        return
    if a.dataset_name != 'Custom_Gestures':
        raise SystemExit('--inference is built for -data Custom_Gestures (main.sh:27)')
    from sais_amd.inference import run_windows, save_inference_outputs, tta_probs
    from sais_amd.model_io import loadModel
    from sais_amd.parallel import init_from_env
    from sais_amd.postprocess import read_frame_counts
    # under torch.distributed.run the window batches of every video are sharded over the ranks and rank 0 writes the files
    # (the reference pins world_size = 1, run_experiments.py:112)
    rank, world, local = init_from_env()
    if world > 1:
        a.local_rank = local
    rgb = load_reps(a.path, '%s_RepsAndLabels' % a.encoder_params)
    flow = load_reps(a.path, 'ViT_SelfSupervised_ImageNet_FlowRepsAndLabels')
    for domain in a.domains:
        for fold in range(a.nfolds):
            savepath = os.path.join(a.path, 'params/Fold_%i' % fold)                 # getSavepath :82-83
            print('***** \n Savepath: %s \n *****' % savepath)
            md, opt, dev = loadModel(a.local_rank, 1, savepath, a.data_type, a.nclasses, domain, a.rep_dim, a.model,
                                     a.task, fold, lr=a.learning_rate, modalities=a.modalities,
                                     freeze_encoder_params=a.freeze_encoder, self_attention=a.self_attention,
                                     importance_loss=a.importance_loss, inference=True)
            for phase in a.phases:
                all_reps, all_attn, all_imp = {"reps": ([], [], []), "labels": [], "videonames": [], "logits": []}, [], []
                # videos and their frame counts come from paths/Custom_Paths.csv, as in the reference
                # (prepare_dataset.py:1705-1727) — NOT from whatever the reps file holds
                counts = read_frame_counts(os.path.join(a.path, 'paths', 'Custom_Paths.csv'))
                for video, total_frames in counts.items():
                    if video not in rgb or video not in flow:
                        raise SystemExit('no features for video %r in the reps files: run extract_representations.py '
                                         'for it first' % video)
                    # the feature files are memory-mapped (read-only views): one video at a time is copied to the GPU
                    x = torch.tensor(rgb[video], dtype=torch.float32).to(dev)
                    f = torch.tensor(flow[video], dtype=torch.float32).to(dev)
                    r, attn, imp = run_windows(md['model'], x, f, videoname=video, batch_size=a.batch_size,
                                               total_frames=total_frames, rank=rank, world_size=world)
                    for v in range(3):
                        all_reps["reps"][v].extend(r["reps"][v])
                    all_reps["labels"] += r["labels"]
                    all_reps["videonames"] += r["videonames"]
                    all_attn += attn
                    all_imp += imp
                if rank == 0:
                    save_inference_outputs(savepath, phase, all_reps, all_attn, all_imp)
                    probs = tta_probs(all_reps, md['prototypes'])
                    print('[%s] %i windows; mean class probabilities %s' % (phase, probs.shape[0], probs.mean(0).tolist()))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        print('Time taken (s): %.3f' % (time.time() - t0))


if __name__ == '__main__':
    main()
