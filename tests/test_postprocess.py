"""Post-processing stage against CSVs written by the reference's own process_inference_results.py
(tests/golden/make_golden_postproc.py ran it, unmodified, on the synthetic project directories whose inputs are
stored next to each CSV)."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from sais_amd import postprocess as pp

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["runs", "noisy", "single", "oneclass", "gaps"]


def _project(tmp_path, case):
    z = np.load(os.path.join(GOLD, "postproc_%s.npz" % case))
    video, total = str(z["video"]), int(z["total_frames"])
    root = str(tmp_path)
    os.makedirs(os.path.join(root, "paths"))
    os.makedirs(os.path.join(root, "params", "Fold_0"))
    with open(os.path.join(root, "paths", "Custom_Paths.csv"), "w") as fh:
        fh.write(",path,category,label\n")
        for i in range(total):
            fh.write("%d,images/%s/frames_%08d.jpg,%s,%s\n" % (i, video, i, video, video))
    reps = z["reps"]
    info = {"reps": tuple([torch.from_numpy(r[i].copy()) for i in range(r.shape[0])] for r in reps),
            "labels": [torch.tensor(0)] * reps.shape[1], "videonames": [video] * reps.shape[1], "logits": []}
    torch.save(info, os.path.join(root, "params", "Fold_0", "reps_and_labels_Custom_inference"))
    pd_ = nn.ParameterDict()
    for c in range(2):
        pd_[str(c)] = nn.Parameter(torch.from_numpy(z["protos"][c:c + 1].copy()))
    torch.save(pd_, os.path.join(root, "params", "Fold_0", "prototypes.zip"))
    return root, z


def _parse(text):
    lines = text.strip().split("\n")
    return lines[0], [ln.split(",") for ln in lines[1:]]


@pytest.mark.parametrize("case", CASES)
def test_csv_matches_reference(tmp_path, case):
    root, _ = _project(tmp_path, case)
    dst, rows = pp.process(root)
    assert dst.endswith("results/Custom_inference_gestures.csv")
    got = open(dst).read()
    want = open(os.path.join(GOLD, "postproc_%s.csv" % case)).read()
    gh, gr = _parse(got)
    wh, wr = _parse(want)
    assert gh == wh and len(gr) == len(wr)
    for g, w in zip(gr, wr):
        assert g[0] == w[0] and g[3:5] == w[3:5] and g[6:] == w[6:]            # index, frames, labels, times, paths
        np.testing.assert_allclose([float(g[i]) for i in (1, 2, 5)], [float(w[i]) for i in (1, 2, 5)],
                                   rtol=0, atol=2e-7)
    assert got == want, "numerically equal but not byte-identical"


def test_group_intervals_edge_cases():
    assert pp.group_intervals([7]) == ([7], [7])
    assert pp.group_intervals([1, 2, 3, 4]) == ([1], [4])
    # the last window joining a group as its 2nd member collapses the group to that window (reference :158-161)
    assert pp.group_intervals([1, 2]) == ([2], [2])
    assert pp.group_intervals([1, 2, 3, 9, 10]) == ([1, 9], [3, 10])
    assert pp.group_intervals([1, 2, 3, 9]) == ([1, 9], [3, 9])
    assert pp.group_intervals([0, 3, 6, 10]) == ([0, 10], [6, 10])             # gap of exactly 3 merges, 4 splits


def test_group_intervals_vs_reference_vectors():
    import json
    vec = json.load(open(os.path.join(GOLD, "postproc_intervals.json")))
    assert len(vec) == 300
    for v in vec:
        assert pp.group_intervals(v["indices"]) == (v["starts"], v["ends"]), v


def test_clock_wraps_like_reference():
    assert pp.frames_to_clock(0) == (0, 0, 0)
    assert pp.frames_to_clock(30 * 3725) == (1, 2, 5)
    with pytest.raises(ValueError):
        pp.frames_to_clock(30 * 3600 * 24)                                     # hour 24 -> strptime failure upstream


def test_length_mismatch_is_loud(tmp_path):
    root, z = _project(tmp_path, "gaps")
    with pytest.raises(ValueError):
        pp.process(root, probs=np.full((3, 2), 0.5, np.float32))


def test_generate_paths_matches_reference_csvs(tmp_path):
    """tests/golden/paths_vidA48/*.csv were written by the reference's generate_paths.py on 48 empty frame files."""
    import subprocess
    import sys
    root = str(tmp_path)
    os.makedirs(os.path.join(root, "images", "vidA"))
    for i in range(48):
        open(os.path.join(root, "images", "vidA", "frames_%08d.jpg" % i), "w").close()
    script = os.path.join(os.path.dirname(GOLD), "..", "SAIS", "scripts", "generate_paths.py")
    subprocess.run([sys.executable, script, "-f", "vidA", "-p", root + "/"], check=True, capture_output=True)
    for name in ("Custom_Paths.csv", "Custom_FlowPaths.csv"):
        assert open(os.path.join(root, "paths", name)).read() == \
            open(os.path.join(GOLD, "paths_vidA48", name)).read()
    assert pp.read_frame_counts(os.path.join(root, "paths", "Custom_Paths.csv")) == {"vidA": 48}


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["noisy", "gaps"])
def test_hip_head_probs_give_same_gestures(tmp_path, case):
    """Probabilities from the HIP head kernel (inference.tta_probs) -> same intervals / labels / times as the
    reference's CSV, probabilities within 1e-6."""
    from sais_amd.inference import tta_probs
    root, z = _project(tmp_path, case)
    dev = torch.device("cuda:0")
    info = {"reps": tuple([torch.from_numpy(r[i].copy()) for i in range(r.shape[0])] for r in z["reps"])}
    protos = nn.ParameterDict()
    for c in range(2):
        protos[str(c)] = nn.Parameter(torch.from_numpy(z["protos"][c:c + 1].copy()).to(dev))
    probs = tta_probs(info, protos).float().cpu().numpy()
    dst, rows = pp.process(root, probs=probs)
    _, gr = _parse(open(dst).read())
    _, wr = _parse(open(os.path.join(GOLD, "postproc_%s.csv" % case)).read())
    assert len(gr) == len(wr)
    for g, w in zip(gr, wr):
        assert g[0] == w[0] and g[3:5] == w[3:5] and g[6:] == w[6:]
        np.testing.assert_allclose([float(g[i]) for i in (1, 2, 5)], [float(w[i]) for i in (1, 2, 5)],
                                   rtol=0, atol=1e-6)
