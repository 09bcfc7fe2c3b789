"""Post-processing of the inference outputs into gesture intervals — the last stage of SAIS/main.sh.

Host logic (the reference runs this stage with device='cpu', process_inference_results.py:204); no kernels here.
The per-window class probabilities can come from two places:
  * `window_probs(reps, prototypes)`     — float32 torch ops on the host, same operation order as the reference's
                                           calcProbs (:76-91) so the CSV is reproducible to the last digit;
  * `sais_amd.inference.tta_probs(...)`  — the HIP head kernel, when the caller is still holding device tensors.

Reference behaviour restated (SAIS/scripts/process_inference_results.py):
  :50-72     windows of 15 frames / hop 15 per video, from the per-video frame count in paths/Custom_Paths.csv
  :105-113   probabilities per TTA version; :218 mean over the 3 versions; :228 mean over folds
  :129-137   entropy = -sum p ln p (float32); window prediction = int(p[1] > 0.515) (:230)
  :233-244   per video and per predicted gesture, keep windows with entropy <= 0.66
  :139-169   groupPredictionIntervals: merge windows whose index gap is <= 3 — including its edge behaviour:
             a group that the LAST window joins as its second member collapses to that last window only
  :171-184   per interval: StartFrame of the first window, EndFrame of the last, mean probability, argmax, entropy
  :186-199   FramesToTime: frame//30 -> (h % 60, m % 60, s % 60) on the date 1900-01-01
  :246-258   CSV columns  ,0,1,StartFrame,EndFrame,Entropy,pred,StartTime,EndTime,Gesture,Video,Path
"""
import csv
import io
import os
from collections import OrderedDict

import numpy as np
import torch

DURATION_FRAMES, HOP_FRAMES, FPS = 15, 15, 30
GESTURES = ("in-view", "out-of-view")          # sorted(['in-view','out-of-view']) -> class 0, class 1 (:74-75)
THRESHOLD = 0.515
SECONDS = 3
ENTROPY_THRESH = 0.66
CSV_COLUMNS = ("", "0", "1", "StartFrame", "EndFrame", "Entropy", "pred", "StartTime", "EndTime", "Gesture", "Video",
               "Path")


# ------------------------------------------------------------------------------------------ inputs
def read_frame_counts(paths_csv):
    """{video: number of frames}, videos in sorted order — the groupby(['category','label']).count() of :52-53."""
    counts = {}
    with open(paths_csv, newline="") as fh:
        rd = csv.reader(fh)
        header = next(rd)
        ic, il = header.index("category"), header.index("label")
        for row in rd:
            key = (row[ic], row[il])
            counts[key] = counts.get(key, 0) + 1
    return OrderedDict((label, n) for (_, label), n in sorted(counts.items()))


def window_table(frame_counts):
    """Per-window (video, StartFrame, EndFrame) arrays, videos concatenated in order (:61-71)."""
    videos, starts, ends = [], [], []
    for video, total in frame_counts.items():
        nsamples = (total - DURATION_FRAMES) // HOP_FRAMES + 1
        for n in range(max(nsamples, 0)):
            videos.append(video)
            starts.append(n * HOP_FRAMES)
            ends.append(n * HOP_FRAMES + DURATION_FRAMES)
    return videos, np.asarray(starts, np.int64), np.asarray(ends, np.int64)


def window_probs(reps, prototypes):
    """softmax over prototypes of the cosine similarity, [n,C] float32 (calcProbs :76-91; no epsilon in the norms)."""
    reps = torch.as_tensor(np.asarray(reps), dtype=torch.float32) if not torch.is_tensor(reps) else reps.float().cpu()
    pros = torch.as_tensor(np.asarray(prototypes), dtype=torch.float32) if not torch.is_tensor(prototypes) \
        else prototypes.float().cpu()
    p_norm = pros / torch.norm(pros, dim=1).unsqueeze(1)
    s_norm = reps / torch.norm(reps, dim=1).unsqueeze(1)
    e = torch.exp(torch.matmul(s_norm, p_norm.T))
    return (e / torch.sum(e, 1).unsqueeze(1)).numpy()


def mean_over_versions(per_version):
    """Mean of equally-shaped float32 arrays the way the groupby mean of :218 / :228 computes it: compensated
    (Kahan) summation carried in float32, then one float32 division by the count."""
    s = np.zeros(np.shape(per_version[0]), np.float32)
    comp = np.zeros_like(s)
    for p in per_version:
        y = np.asarray(p, np.float32) - comp
        t = s + y
        comp = (t - s) - y
        s = t
    return s / np.float32(len(per_version))


def load_fold_probs(savepath, inference_set="Custom_inference"):
    """TTA-averaged probabilities of one fold directory (getResults :99-113 + :218)."""
    info = torch.load(os.path.join(savepath, "reps_and_labels_%s" % inference_set), map_location="cpu",
                      weights_only=False)
    protos = torch.load(os.path.join(savepath, "prototypes.zip"), map_location="cpu", weights_only=False)
    pros = torch.vstack([p.detach() for p in protos.values()])
    reps = info["reps"]
    if not isinstance(reps, tuple):
        raise ValueError("reps_and_labels_%s was written without test-time augmentation" % inference_set)
    return mean_over_versions([window_probs(torch.stack(list(r)).detach(), pros) for r in reps])


# ------------------------------------------------------------------------------------------ decisions
def entropy_f32(probs):
    """-sum_c p ln p per row, float32 arithmetic in the reference's order (:130)."""
    p = np.asarray(probs, np.float32)
    t = p * np.log(p)
    acc = t[:, 0].copy()
    for c in range(1, t.shape[1]):
        acc = acc + t[:, c]
    return -acc


def window_predictions(probs, threshold=THRESHOLD):
    """Class index per window: last class iff its probability exceeds the threshold (:134), argmax if None (:132)."""
    p = np.asarray(probs, np.float32)
    if threshold is None:
        return np.argmax(p, axis=1)
    return (p[:, -1] > threshold).astype(np.int64)


def group_intervals(indices, seconds=SECONDS):
    """Merge sorted window indices into (first, last) pairs; a gap > `seconds` starts a new group (:139-169)."""
    idx = [int(i) for i in indices]
    starts, ends = [], []
    if len(idx) == 1:
        return [idx[0]], [idx[0]]
    start = prev = idx[0]
    since_reset = 0
    for i in idx[1:]:
        if i - prev > seconds:
            starts.append(start)
            ends.append(prev)
            start, since_reset = i, 0
        if i == idx[-1]:
            if since_reset == 0:          # the reference's "final single entry" branch also fires for a 2-member tail
                starts.append(i)
                ends.append(i)
            else:
                starts.append(start)
                ends.append(i)
        since_reset += 1
        prev = i
    return starts, ends


def frames_to_clock(frame, fps=FPS):
    """(h, m, s) as FramesToTime computes them (:186-196): each of hours/minutes/seconds taken modulo 60."""
    sec = int(frame) // fps
    mins = sec // 60
    hours = mins // 60
    h, m, s = hours % 60, mins % 60, sec % 60
    if h > 23:
        raise ValueError("time data %d-%d-%d does not match format '%%H-%%M-%%S'" % (h, m, s))
    return h, m, s


def _mean_f32(col):
    col = np.ascontiguousarray(col, np.float32)
    return np.float32(np.sum(col) / np.float32(col.shape[0]))


def gesture_intervals(videos, start_frames, end_frames, probs, threshold=THRESHOLD, seconds=SECONDS,
                      entropy_thresh=ENTROPY_THRESH):
    """The rows of Custom_inference_gestures.csv as a list of dicts (:230-252)."""
    probs = np.asarray(probs, np.float32)
    ent = entropy_f32(probs)
    pred = window_predictions(probs, threshold)
    videos = np.asarray(videos, dtype=object)
    rows = []
    for video in list(OrderedDict.fromkeys(videos.tolist())):
        in_video = videos == video
        for cls, gesture in enumerate(GESTURES):
            keep = np.nonzero(in_video & (pred == cls) & (ent <= np.float32(entropy_thresh)))[0]
            if keep.size == 0:
                continue
            firsts, lasts = group_intervals(keep, seconds)
            for k, (a, b) in enumerate(zip(firsts, lasts)):
                member = keep[(keep >= a) & (keep <= b)]
                mean = np.array([_mean_f32(probs[member, c]) for c in range(probs.shape[1])], np.float32)
                rows.append({"index": k, "probs": mean, "StartFrame": int(start_frames[a]),
                             "EndFrame": int(end_frames[b]), "Entropy": entropy_f32(mean[None])[0],
                             "pred": GESTURES[int(np.argmax(mean))],
                             "StartTime": frames_to_clock(start_frames[a]), "EndTime": frames_to_clock(end_frames[b]),
                             "Gesture": gesture, "Video": video, "Path": os.path.join("images", video)})
    return rows


# ------------------------------------------------------------------------------------------ output
def _clock_column(clocks):
    # a datetime column whose every entry is midnight is written as the bare date
    if all(c == (0, 0, 0) for c in clocks):
        return ["1900-01-01"] * len(clocks)
    return ["1900-01-01 %02d:%02d:%02d" % c for c in clocks]


def format_csv(rows):
    """CSV text with the reference's columns and number formatting (float32 shortest round-trip repr)."""
    out = io.StringIO()
    w = csv.writer(out, lineterminator="\n")
    if not rows:                       # an empty frame gets a Path column first (:254) -> KeyError in the reference
        raise KeyError("Video")
    w.writerow(CSV_COLUMNS)
    st = _clock_column([r["StartTime"] for r in rows])
    et = _clock_column([r["EndTime"] for r in rows])
    for r, s, e in zip(rows, st, et):
        w.writerow([r["index"]] + [str(np.float32(p)) for p in r["probs"]] +
                   [r["StartFrame"], r["EndFrame"], str(np.float32(r["Entropy"])), r["pred"], s, e, r["Gesture"],
                    r["Video"], r["Path"]])
    return out.getvalue()


def process(rootpath, folds=(0,), inference_set="Custom_inference", probs=None):
    """paths/Custom_Paths.csv + params/Fold_k/{reps_and_labels_<set>, prototypes.zip} -> results/<set>_gestures.csv.
    `probs` ([n,2] float32) overrides the per-fold files, e.g. with the output of inference.tta_probs."""
    videos, sf, ef = window_table(read_frame_counts(os.path.join(rootpath, "paths", "Custom_Paths.csv")))
    if probs is None:
        per_fold = [load_fold_probs(os.path.join(rootpath, "params/Fold_%i" % f), inference_set) for f in folds]
        probs = mean_over_versions(per_fold)
    probs = np.asarray(probs, np.float32)
    if probs.shape[0] != len(videos):
        raise ValueError("Length mismatch: %d windows in paths/Custom_Paths.csv, %d in the inference outputs"
                         % (len(videos), probs.shape[0]))
    rows = gesture_intervals(videos, sf, ef, probs)
    text = format_csv(rows)
    os.makedirs(os.path.join(rootpath, "results"), exist_ok=True)
    dst = os.path.join(rootpath, "results", "%s_gestures.csv" % inference_set)
    with open(dst, "w", newline="") as fh:
        fh.write(text)
    return dst, rows
