import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def pytest_configure(config):
    # The CPU oracle (fp32 torch) is the slow half of the parity tests at benchmark sizes.  Its GEMMs stop scaling past a few dozen
    # threads and get much slower when torch takes every core of a large host (bench.py's cpu_baseline leg caps them the same way).
    import torch
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(32, avail)))
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(32, avail))))     # the CLI tests' child processes inherit it
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "experimental: rejected kernel forms, built only with -DSAIS_EXPERIMENTAL=1 (not part of -m gpu)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    return load


# ---- worst observed deviation per parity quantity of a run (report the values, not just pass / fail).
# Tests call parity.parity_log(name, value, bar); the session writes gpurun_out/parity_worst.json (max per name).
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_sessionfinish(session, exitstatus):
    import parity
    if not parity.WORST:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_worst.json"), "w") as fh:
        json.dump(dict(sorted(parity.WORST.items())), fh, indent=1)
