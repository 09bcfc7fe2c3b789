"""Training harness of the SAIS hot path on MI355X — drop-in for the reference's
SAIS/scripts/train.py:18-121 (`trainModel`), perform_training.py:49-227 (`single_epoch`),
prepare_dataset.py:2798-2899 (`createPaddingMask`, `pad_collate`, Prototypes branch) and
prepare_miscellaneous.py:111-185 (`calcNCEMetrics`), for `-t Prototypes -m ViT -dt reps`.

What runs where: the arithmetic of a step (temporal encoder forward / backward, prototype loss, SGD) is the HIP
path (sais_amd.temporal / loss / optim); everything in this file is the reference's HOST logic restated — batching,
padding, the epoch / phase loop, early stopping on the validation loss (patience 5), the best-validation snapshot and
the files rank 0 writes: `params` (module.-prefixed state_dict), `prototypes` (pickled nn.ParameterDict), `metrics`,
`reps_and_labels` — which `loadModel(..., inference=True)` reads back once renamed *.zip (README.md:64-75).

Dataset.  The reference trains on private surgical datasets through prepare_dataset.VideoDataset; what a Prototypes /
reps item IS there (prepare_dataset.py:2631-2700) is restated by `GestureWindows`: annotated windows
(Video, Gesture, StartFrame, EndFrame) over the per-video feature files, `arange(start-1, end-1, (end-start)//10)` frame
indices (+3 / +6 offsets for the two extra test-time-augmentation versions in val / test / inference), flow rows
`unique(idx // 15)`.  The annotation table is this build's own CSV (paths/<dataset>_Annotations.csv: Video, Gesture,
StartFrame, EndFrame, phase) since the reference's tables are not public.

Data parallel: with torch.distributed initialised, every rank takes a strided shard of the training windows, the
gradients are exchanged by sais_amd.parallel.GradSync (touched slices only) and 1/world is folded into SGD.
"""
import copy
import csv
import os
from collections import defaultdict

import numpy as np
import torch

from . import _lib as L
from .loss import calcImportanceLoss, calcNCELoss, cosine_logits_and_probs
from .model_io import loadModel, save_prototypes_file

FLOW_JUMP = 15
TTA_OFFSETS = (0, 3, 6)


# ------------------------------------------------------------------------------------------ collate (host)
def pad_stream(seqs):
    """Zero-pad a list of [nsnippets, T_b, ...] tensors along T to the batch maximum.
    Returns (padded [B, nsnippets, Tmax, ...], key_padding_mask bool [B, nsnippets, Tmax + 1], lens).  Mask slot 0 is the
    CLS token the temporal encoder prepends, slots lens[b] + 1 .. are the padding keys (True = masked): the contract of
    the reference's createPaddingMask (prepare_dataset.py:2798-2806)."""
    lens = [int(t.shape[1]) for t in seqs]
    ns, tmax = max(int(t.shape[0]) for t in seqs), max(lens)
    padded = seqs[0].new_zeros((len(seqs), ns, tmax) + tuple(seqs[0].shape[2:]))
    mask = torch.arange(tmax + 1).expand(len(seqs), ns, tmax + 1) > torch.tensor(lens).view(-1, 1, 1)
    for b, t in enumerate(seqs):
        padded[b, :, :lens[b]] = t            # a clip with another snippet count than the batch's raises here
    return padded, mask, lens


def createPaddingMask(x, lens):
    """Reference signature (prepare_dataset.py:2798): x = list of [nframes, nsnippets, dim] tensors."""
    ns = max(int(t.shape[1]) for t in x)
    return torch.arange(max(lens) + 1).expand(len(x), ns, max(lens) + 1) > torch.tensor(list(lens)).view(-1, 1, 1)


def pad_collate(batch):
    """Collate of the Prototypes path (contract: prepare_dataset.py:2839-2899).  Items are (videoname, snippets, flows,
    label, frames_importance, domain); snippets / flows are [nsnippets, T, dim] tensors (training) or tuples with one
    such tensor per test-time-augmentation version.  Returns the 11-tuple the epoch loop unpacks: names, RGB batch, flow
    batch, importance targets, labels, RGB lens, flow lens, RGB mask, flow mask, importance mask, domains — under TTA the
    batch / lens / mask entries are dicts keyed by the version number and the importance entries are placeholders."""
    names, rgb, flow, labels, importance, domains = zip(*batch)
    if isinstance(rgb[0], tuple):
        cols = {k: {} for k in ("xp", "xm", "xl", "fp", "fm", "fl")}
        for v in range(len(rgb[0])):
            cols["xp"][v], cols["xm"][v], cols["xl"][v] = pad_stream([item[v] for item in rgb])
            cols["fp"][v], cols["fm"][v], cols["fl"][v] = pad_stream([item[v] for item in flow])
        xp, xm, xl, fp, fm, fl = (cols[k] for k in ("xp", "xm", "xl", "fp", "fm", "fl"))
        ip, im = torch.zeros(1, 1), torch.zeros(1, 1)
    else:
        xp, xm, xl = pad_stream(list(rgb))
        fp, fm, fl = pad_stream(list(flow))
        ip, im, _ = pad_stream(list(importance))          # targets [1, T] per clip -> [B, 1, Tmax], mask [B, 1, Tmax + 1]
    return list(names), xp, fp, ip, torch.stack(list(labels)), xl, fl, xm, fm, im, domains


# ------------------------------------------------------------------------------------------ dataset (host)
class GestureWindows(torch.utils.data.Dataset):
    """Prototypes / reps items as prepare_dataset.VideoDataset.__getitem__ builds them (:2631-2700)."""

    def __init__(self, rows, rgb_reps, flow_reps, classes, phase, domain):
        self.rows, self.rgb, self.flow = rows, rgb_reps, flow_reps
        self.classes = sorted(classes)                         # LabelEncoder order
        self.phase, self.domain = phase, domain

    def __len__(self):
        return len(self.rows)

    def _version(self, video_reps, flow_reps, start, end, off, jump):
        idx = list(np.arange(start + off, end, jump))
        x = torch.tensor(video_reps[idx, :], dtype=torch.float).unsqueeze(0)
        rows = [r for r in np.unique([i // FLOW_JUMP for i in idx]) if r < len(flow_reps)]
        f = torch.tensor(flow_reps[rows, :], dtype=torch.float).unsqueeze(0)
        return x, f

    def __getitem__(self, i):
        r = self.rows[i]
        video = r["Video"]
        label = torch.tensor(self.classes.index(r["Gesture"]), dtype=torch.long)
        start, end = int(r["StartFrame"]) - 1, int(r["EndFrame"]) - 1
        jump = max((end - start) // 10, 1)
        video_reps, flow_reps = self.rgb[video], self.flow[video]
        if self.phase in ("train", "train+val"):
            x, f = self._version(video_reps, flow_reps, start, end, 0, jump)
            return video, x, f, label, torch.zeros(1, x.shape[1], dtype=torch.float), self.domain
        vs = [self._version(video_reps, flow_reps, start, end, off, jump) for off in TTA_OFFSETS]
        xs, fs = tuple(v[0] for v in vs), tuple(v[1] for v in vs)
        return video, xs, fs, label, torch.zeros(1, xs[0].shape[1], dtype=torch.float), self.domain


def read_annotations(path):
    """paths/<dataset>_Annotations.csv -> {phase: [row dict]} and the sorted class list."""
    by_phase, classes = defaultdict(list), set()
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            by_phase[row["phase"]].append(row)
            classes.add(row["Gesture"])
    return by_phase, sorted(classes)


def load_dataloaders(root_path, dataset_name, batch_size, phases, domain, encoder_params, rank=0, world_size=1, seed=0):
    """loadDataloader(...).load() (:2790-2796): shuffle only the training phases, drop_last=False, pad_collate."""
    from .hdf5_min import read_h5
    rgb = read_h5(os.path.join(root_path, "results", "%s_RepsAndLabels.h5" % encoder_params))
    flow = read_h5(os.path.join(root_path, "results", "ViT_SelfSupervised_ImageNet_FlowRepsAndLabels.h5"))
    by_phase, classes = read_annotations(os.path.join(root_path, "paths", "%s_Annotations.csv" % dataset_name))
    loaders = {}
    for phase in phases:
        rows = by_phase[phase]
        train = phase in ("train", "train+val")
        if train and world_size > 1:
            # data parallel: a strided shard of the windows per rank, padded by wrapping around (DistributedSampler's
            # rule) so that every rank runs the same number of batches — the gradient all-reduces are collective
            total = -(-len(rows) // world_size) * world_size
            rows = (rows + rows[:total - len(rows)])[rank::world_size]
        ds = GestureWindows(rows, rgb, flow, classes, phase, domain)
        g = torch.Generator().manual_seed(seed)
        loaders[phase] = torch.utils.data.DataLoader(ds, batch_size=batch_size, shuffle=train, drop_last=False,
                                                     collate_fn=pad_collate, generator=g)
    return loaders, classes


# ------------------------------------------------------------------------------------------ metrics (host)
def calcNCEMetrics(rank, snip_sequence_list, labels_list, videoname_list, gesture_prototypes):
    """prepare_miscellaneous.py:111-171: probabilities (mean over the TTA versions), accuracy, macro precision / recall,
    AUC.  The probabilities come from the HIP head kernel; the scores are sklearn on the host, as in the reference."""
    from sklearn.metrics import precision_score, recall_score, roc_auc_score
    keys = list(gesture_prototypes.keys())
    labels = torch.stack([l.detach().cpu() for l in labels_list])
    cols = torch.tensor([keys.index(str(int(l))) for l in labels])
    versions = snip_sequence_list if isinstance(snip_sequence_list, tuple) else (snip_sequence_list,)
    probs = None
    for v in versions:
        p = cosine_logits_and_probs(torch.stack(list(v)), gesture_prototypes)[1].cpu()
        probs = p if probs is None else probs + p
    probs = probs / len(versions)
    preds = torch.argmax(probs, 1)
    acc = (torch.sum(preds == cols) / preds.shape[0]).item()
    y, yhat, pr = cols.numpy(), preds.numpy(), probs.numpy()
    prec = precision_score(y, yhat, average="macro", zero_division=0)
    rec = recall_score(y, yhat, average="macro", zero_division=0)
    if len(keys) == 2:
        pr = pr[:, -1]
    try:
        auc = roc_auc_score(y, pr, multi_class="ovr")
    except Exception:
        auc = float("nan")
    return acc, auc, prec, rec


def trackMetrics(metrics, metrics_dict):
    for name, value in metrics.items():
        metrics_dict[name].append(value)
    return metrics_dict


def printMetrics(phase, metrics):
    names = [phase + "_" + n for n in metrics]
    print("  ".join(names))
    print("  ".join("%.3f" % v for v in metrics.values()))


# ------------------------------------------------------------------------------------------ one epoch
def single_epoch(rank, world_size, dataloader, model_dict, optimizer, device, phase, nclasses, task, importance_loss,
                 sync=None):
    """perform_training.single_epoch (:49-227), Prototypes task.  Returns (metrics, snip_sequence_list, labels_list,
    videoname_list, attention_list, importance_list, output_logits_list) with the reference's structure."""
    if task != "Prototypes":
        raise NotImplementedError("only task 'Prototypes' is on the MI355X hot path")
    model = model_dict["model"]
    lists = ([], [], [])
    attention_list, importance_list, labels_list, videoname_list = [], [], [], []
    running_loss, nitems, is_list = 0.0, 0, False
    for (videoname, snippets, flows, importances, labels, xlens, flens, xpad, fpad, ipad, domains) in dataloader[phase]:
        if isinstance(snippets, dict):                                         # TTA versions (:92-100)
            snippets = [s.to(device) for s in snippets.values()]
            xpad = [m.to(device) for m in xpad.values()]
            xlens = [l for l in xlens.values()]
            flows = [f.to(device) for f in flows.values()]
            fpad = [m.to(device) for m in fpad.values()]
            flens = [l for l in flens.values()]
        else:
            snippets, xpad, fpad, flows = snippets.to(device), xpad.to(device), fpad.to(device), flows.to(device)
        with torch.set_grad_enabled(phase == "train"):
            if importance_loss:
                output_importances, snip_sequence, snip_attn = model(snippets, flows, xlens, flens, task, xpad, fpad, domains)
            else:
                snip_sequence, snip_attn = model(snippets, flows, xlens, flens, task, xpad, fpad, domains)
            is_list = isinstance(snip_sequence, list)
            if "inference" in phase:
                loss = torch.tensor(0)                                         # :121-122
            elif is_list:
                loss = torch.mean(torch.stack([calcNCELoss(rank, s, labels, videoname, model_dict["prototypes"], domains).detach()
                                               for s in snip_sequence]))
            else:
                loss = calcNCELoss(rank, snip_sequence, labels, videoname, model_dict["prototypes"], domains)
                if phase == "train" and importance_loss:
                    loss = loss + calcImportanceLoss(output_importances, importances, ipad, labels)
        if phase == "train":
            optimizer.zero_grad()
            loss.backward()
            if sync is not None:
                sync.reduce_params(model_dict["prototypes"].values())
                sync.wait()
            optimizer.step(grad_scale=1.0 / world_size)
        if is_list:
            for v in range(3):
                lists[v].extend(s.detach() for s in snip_sequence[v])
        else:
            lists[0].extend(s.detach() for s in snip_sequence)
        attention_list.append(snip_attn.detach())
        labels_list.extend(labels)
        videoname_list.extend(videoname)
        if importance_loss:
            xl = xlens[0] if is_list else xlens
            importance_list.append([imp[:, 1:n + 1, :].squeeze() for imp, n in zip(output_importances.detach(), xl)])
        bsz = snippets[0].shape[0] if is_list else snippets.shape[0]
        running_loss += float(loss.detach()) * bsz
        nitems += bsz
    ave_loss = running_loss / max(len(dataloader[phase].dataset), 1)
    snip_sequence_list = lists if is_list else lists[0]
    if phase == "inference" or nitems == 0:
        acc, auc, prec, rec = 0, 0, 0, 0
    else:
        acc, auc, prec, rec = calcNCEMetrics(rank, snip_sequence_list, labels_list, videoname_list, model_dict["prototypes"])
    metrics = {"loss": ave_loss, "acc": acc, "auc": auc, "precision": prec, "recall": rec}
    return metrics, snip_sequence_list, labels_list, videoname_list, attention_list, importance_list, []


# ------------------------------------------------------------------------------------------ training loop
def trainModel(rank, world_size, root_path, savepath, dataset_name, data_type, batch_size, nclasses, domain, phases, lr,
               modalities, freeze_encoder_params, inference, task, balance, balance_groups, single_group, group_info,
               self_attention, importance_loss, encoder_type, encoder_params, snippetLength, frameSkip, overlap, rep_dim,
               nepochs, fold, training_fraction, dataloader=None):
    """train.py:18-121 with the reference's positional signature.  `dataloader` ({phase: DataLoader}) overrides the
    files under root_path.  Returns the metrics history {name: [per validation epoch]}."""
    if not torch.cuda.is_available():
        raise L.SaisHipError("trainModel needs an MI355X: the HIP path has no CPU fallback")
    import torch.distributed as dist
    from .parallel import GradSync
    model, optimizer, device = loadModel(rank, world_size, savepath, data_type, nclasses, domain, rep_dim, encoder_type,
                                         task, fold, lr=lr, modalities=modalities,
                                         freeze_encoder_params=freeze_encoder_params, self_attention=self_attention,
                                         importance_loss=importance_loss, inference=inference)
    model["model"].dropout_seed = 1000 * fold + rank       # train-mode dropout: every fold and rank its own mask stream
    if dataloader is None:
        dataloader, _ = load_dataloaders(root_path, dataset_name, batch_size, phases, domain, encoder_params, rank,
                                         world_size, seed=fold)
    sync = None
    if world_size > 1 and dist.is_initialized():
        sync = GradSync(world_size)
        tm = model["model"]
        tm._engine(device)
        # loadModel draws the initial weights and prototypes from every process's own RNG (prepare_model.py:556-560 runs in
        # one process): replicas must start from rank 0's, or the averaged gradients are applied to different parameters
        with torch.no_grad():
            sync.broadcast_initial_state([tm.flat.flat] + [p.data for p in model["prototypes"].values()])
        tm._sig = None                                        # the bf16 / transposed shadows follow at the next forward
        # longest window of any rank's shard: the position rows 0 .. T-1 are the only ones that receive gradients
        tmax = max((_longest_window(dataloader[ph].dataset) for ph in phases if ph in ("train", "train+val")), default=0)
        tmax = sync.agree_max(tmax, device)
        tm.grad_ready_hook = sync.temporal_hook(tm, tmax if tmax > 0 else None)
    best_params_dict, best_prototypes_dict, reps_and_labels_dict = {}, {}, {}
    attention_dict, importance_dict = [], []
    metrics_dict = defaultdict(list)
    min_loss, epoch_count, max_patience, patience_count = float("inf"), 1, 5, 1
    while epoch_count <= nepochs and patience_count <= max_patience:
        print("\n **** Epoch %i ****" % epoch_count)
        for phase in phases:
            print(phase)
            if phase == "train" and not inference:
                model["model"].train()
            else:
                model["model"].eval()
            metrics, snippets, labels, videonames, attention, importance, logits = single_epoch(
                rank, world_size, dataloader, model, optimizer, device, phase, nclasses, task, importance_loss, sync)
            printMetrics(phase, metrics)
            if not inference:
                if phase == "val":
                    loss = metrics["loss"]
                    if sync is not None:                                   # one stop decision for all ranks
                        loss = metrics["loss"] = sync.mean_scalar(loss, device)
                    metrics_dict = trackMetrics(metrics, metrics_dict)
                    if loss < min_loss:
                        min_loss, patience_count = loss, 1
                        best_params_dict = {k: v.detach().cpu().clone() for k, v in model["model"].state_dict().items()}
                        reps_and_labels_dict = {"reps": _to_cpu(snippets), "labels": [l.cpu() for l in labels],
                                                "videonames": videonames, "logits": logits}
                        best_prototypes_dict = copy.deepcopy({k: v.detach().cpu() for k, v in model["prototypes"].items()})
                    else:
                        patience_count += 1
            else:
                reps_and_labels_dict = {"reps": _to_cpu(snippets), "labels": [l.cpu() for l in labels],
                                        "videonames": videonames, "logits": logits}
                attention_dict = [a.cpu() for a in attention]
                importance_dict = importance
        epoch_count += 1
    if rank == 0:                                                          # :98-121
        os.makedirs(savepath, exist_ok=True)
        if not inference:
            if not best_params_dict:                                       # no 'val' phase: keep the final weights
                best_params_dict = {k: v.detach().cpu().clone() for k, v in model["model"].state_dict().items()}
                best_prototypes_dict = {k: v.detach().cpu() for k, v in model["prototypes"].items()}
            # keys carry the `module.` prefix loadModel(inference=True) strips (prepare_model.py:525-527)
            torch.save({"module." + k: v for k, v in best_params_dict.items()}, os.path.join(savepath, "params"))
            torch.save(dict(metrics_dict), os.path.join(savepath, "metrics"))
            save_prototypes_file(best_prototypes_dict, os.path.join(savepath, "prototypes"))
            torch.save(reps_and_labels_dict, os.path.join(savepath, "reps_and_labels"))
            print("All Info Saved!")
        else:
            torch.save(reps_and_labels_dict, os.path.join(savepath, "reps_and_labels_%s" % phases[0]))
            torch.save(attention_dict, os.path.join(savepath, "attention_%s" % phases[0]))
            torch.save(importance_dict, os.path.join(savepath, "importance_%s" % phases[0]))
    return dict(metrics_dict)


def _longest_window(dataset):
    """Frames in the longest training window of a GestureWindows dataset (0 = unknown dataset type)."""
    if not isinstance(dataset, GestureWindows):
        return 0
    best = 0
    for r in dataset.rows:
        start, end = int(r["StartFrame"]) - 1, int(r["EndFrame"]) - 1
        best = max(best, len(np.arange(start, end, max((end - start) // 10, 1))))
    return best


def _to_cpu(snippets):
    if isinstance(snippets, tuple):
        return tuple([s.cpu() for s in v] for v in snippets)
    return [s.cpu() for s in snippets]
