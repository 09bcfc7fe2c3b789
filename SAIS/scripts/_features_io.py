"""Feature-file I/O for the CLI.  The reference stores one [nframes,384] fp32 dataset per video label in
results/<enc>_RepsAndLabels.h5 (extract_representations.saveH5 :389-407).  h5py is used when importable; this
image does not ship it, so the same mapping is then kept in an .npz next to where the .h5 would be."""
import os

import numpy as np


def _paths(root, name):
    base = os.path.join(root, 'results', name)
    return base + '.h5', base + '.npz'


def save_reps(root, name, reps_by_video):
    h5, npz = _paths(root, name)
    os.makedirs(os.path.dirname(h5), exist_ok=True)
    try:
        import h5py
        with h5py.File(h5, 'a') as f:
            for k, v in reps_by_video.items():
                if k in f:
                    del f[k]
                f.create_dataset(k, data=np.asarray(v, dtype=np.float32))
        return h5
    except ImportError:
        old = dict(np.load(npz)) if os.path.exists(npz) else {}
        old.update({k: np.asarray(v, dtype=np.float32) for k, v in reps_by_video.items()})
        np.savez(npz, **old)
        return npz


def load_reps(root, name):
    h5, npz = _paths(root, name)
    if os.path.exists(h5):
        import h5py
        with h5py.File(h5, 'r') as f:
            return {k: np.array(f.get(k)) for k in f.keys()}
    if os.path.exists(npz):
        return dict(np.load(npz))
    raise FileNotFoundError(f"neither {h5} nor {npz} exists: run extract_representations.py first")
