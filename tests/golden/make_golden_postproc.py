#!/usr/bin/env python3
"""Golden vectors for the post-processing stage, produced by RUNNING the reference's
SAIS/scripts/process_inference_results.py (as a subprocess, unmodified, where it lies under /root/reference)
on a synthetic project directory.  Build-container only; writes tests/golden/postproc_<case>.npz + .csv.

    python tests/golden/make_golden_postproc.py

Inputs laid out the way the reference expects them (process_inference_results.py:50,101-102):
  <root>/paths/Custom_Paths.csv                       one row per frame: path, category, label
  <root>/params/Fold_0/reps_and_labels_Custom_inference   {'reps': (list,list,list) of [256] tensors, ...}
  <root>/params/Fold_0/prototypes.zip                 pickled nn.ParameterDict {'0','1'} of [1,256]
Output: <root>/results/Custom_inference_gestures.csv  (kept byte-for-byte as postproc_<case>.csv)
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np
import pandas as pd
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SCRIPT = "/root/reference/SAIS/scripts/process_inference_results.py"

# (name, total frames, seed, structure) — 'structure' shapes how the class signal moves over time so that the
# cases cover: long runs, isolated single windows, gaps of exactly / just above `seconds`=3 index steps, windows
# removed by the entropy gate, a video with a single window, and a video where one class never survives.
CASES = [
    ("runs", 1815, 0, "runs"),
    ("noisy", 2400, 1, "noisy"),
    ("single", 15, 2, "runs"),
    ("oneclass", 600, 3, "oneclass"),
    ("gaps", 930, 4, "gaps"),
]


def synth_reps(nwin, seed, structure):
    g = np.random.default_rng(1000 + seed)
    protos = g.standard_normal((2, 256)).astype(np.float32)
    t = np.arange(nwin)
    if structure == "runs":
        sig = np.sign(np.sin(t / 7.0 + 0.3)) * 0.9
    elif structure == "noisy":
        sig = g.uniform(-1, 1, nwin)
    elif structure == "oneclass":
        sig = np.full(nwin, 0.8)
    else:  # gaps: class-1 islands separated by 1..5 windows of class 0, some uncertain
        sig = np.full(nwin, -0.8)
        pos, k = 2, 0
        while pos < nwin:
            sig[pos:pos + 2] = 0.8
            pos += 2 + 1 + (k % 5)
            k += 1
        sig[::11] = 0.02
    reps = []
    for v in range(3):  # three TTA versions: same signal, different noise
        w = (sig[:, None] * (protos[1] - protos[0])[None] +
             0.35 * g.standard_normal((nwin, 256))).astype(np.float32)
        reps.append(w)
    return reps, protos


def interval_vectors():
    """Random sorted index sets through the reference's own groupPredictionIntervals (:139-169; the function is
    compiled out of the reference file in memory, the file's __main__ part is not run)."""
    import ast
    import json
    tree = ast.parse(open(REF_SCRIPT).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "groupPredictionIntervals"]
    ns = {"pd": pd, "np": np}
    exec(compile(ast.Module(body=fn, type_ignores=[]), REF_SCRIPT, "exec"), ns)
    g = np.random.default_rng(7)
    vec = []
    for trial in range(300):
        n = int(g.integers(1, 14))
        idx = sorted(set(int(v) for v in g.integers(0, 40, n)))
        frame = pd.DataFrame({"x": np.zeros(len(idx))}, index=idx)
        s, e = ns["groupPredictionIntervals"](frame, 3)
        vec.append({"indices": idx, "starts": [int(v) for v in s], "ends": [int(v) for v in e]})
    with open(os.path.join(HERE, "postproc_intervals.json"), "w") as fh:
        json.dump(vec, fh, separators=(",", ":"))
    print("interval vectors", len(vec))


def main():
    interval_vectors()
    for name, total, seed, structure in CASES:
        root = tempfile.mkdtemp(prefix="sais_pp_")
        try:
            os.makedirs(os.path.join(root, "paths"))
            os.makedirs(os.path.join(root, "params", "Fold_0"))
            os.makedirs(os.path.join(root, "results"))  # the reference's own mkdir call is misspelt (:256)
            video = "video_%s" % name
            paths = [os.path.join("images", video, "frames_%08d.jpg" % i) for i in range(total)]
            pd.DataFrame({"path": paths, "category": video, "label": video}).to_csv(
                os.path.join(root, "paths", "Custom_Paths.csv"))
            nwin = (total - 15) // 15 + 1
            reps, protos = synth_reps(nwin, seed, structure)
            info = {"reps": tuple([torch.from_numpy(r[i].copy()) for i in range(nwin)] for r in reps),
                    "labels": [torch.tensor(0)] * nwin, "videonames": [video] * nwin, "logits": []}
            torch.save(info, os.path.join(root, "params", "Fold_0", "reps_and_labels_Custom_inference"))
            pd_ = nn.ParameterDict()
            for c in range(2):
                pd_[str(c)] = nn.Parameter(torch.from_numpy(protos[c:c + 1].copy()))
            torch.save(pd_, os.path.join(root, "params", "Fold_0", "prototypes.zip"))
            # torch>=2.6 defaults weights_only=True; the reference predates it -> flip the default from outside
            env = dict(os.environ, TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD="1")
            r = subprocess.run([sys.executable, REF_SCRIPT, "-p", root], env=env, capture_output=True, text=True)
            if r.returncode != 0:
                sys.stderr.write(r.stdout[-2000:] + r.stderr[-4000:])
                raise SystemExit("reference post-processing failed on case %s" % name)
            shutil.copy(os.path.join(root, "results", "Custom_inference_gestures.csv"),
                        os.path.join(HERE, "postproc_%s.csv" % name))
            np.savez_compressed(os.path.join(HERE, "postproc_%s.npz" % name),
                                reps=np.stack(reps), protos=protos, total_frames=np.int64(total),
                                video=np.array(video))
            print(name, "windows", nwin, "rows",
                  len(pd.read_csv(os.path.join(HERE, "postproc_%s.csv" % name))))
        finally:
            shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
