"""Feature-file I/O for the CLI.  The reference stores one [nframes,384] fp32 dataset per video label in
results/<enc>_RepsAndLabels.h5 (extract_representations.saveH5 :389-407, `h5py.File(path, 'w')`: the file is
TRUNCATED on every run) and reads it with h5py (prepare_dataset.py:1702-1703).  Same files here, written and read by
sais_amd.hdf5_min (no h5py in this image; its output opens in h5py / libhdf5, and it reads what they write)."""
import os

import numpy as np

from sais_amd.hdf5_min import read_h5, write_h5


def reps_path(root, name):
    return os.path.join(root, 'results', name + '.h5')


def save_reps(root, name, reps_by_video):
    path = reps_path(root, name)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    return write_h5(path, {k: np.asarray(v, dtype=np.float32) for k, v in reps_by_video.items()})


def load_reps(root, name):
    path = reps_path(root, name)
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} does not exist: run extract_representations.py first")
    return read_h5(path)
