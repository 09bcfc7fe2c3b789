#!/usr/bin/env python3
"""Pins the window sampler with the reference's OWN dataset code: builds a prepare_dataset.VideoDataset
without its constructor (which opens HDF5 files and private CSVs), hands it in-memory feature tables and the
Custom_inference window table of prepare_dataset.py:1705-1727, and calls its __getitem__ (:1747-2700) for every window.
Features are arange rows, so the returned tensors ARE the frame / flow-row indices.  Writes tests/golden/sampler.npz.

    python tests/golden/make_golden_sampler.py        (build container only: needs /root/reference)"""
import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402

CASES = {"n512": 512, "n77": 77, "n45": 45, "n15": 15}


def main():
    G.import_reference()
    import prepare_dataset as PD
    g = {}
    for name, n in CASES.items():
        ds = PD.VideoDataset.__new__(PD.VideoDataset)
        ds.dataset_name, ds.data_type, ds.phase, ds.domain, ds.task = 'Custom_Gestures', 'reps', 'Custom_inference', 'in_vs_out', 'Prototypes'
        ds.encoder_type, ds.importance_loss = 'ViT', False
        nflow = max(n // 15, 1)
        ds.hf_rgb = {'vid': np.arange(n, dtype=np.float32)[:, None].repeat(384, 1)}
        ds.hf_of = {'vid': np.arange(nflow, dtype=np.float32)[:, None].repeat(384, 1)}
        nsamples = (n - 15) // 15 + 1                                   # :1716-1719
        df = pd.DataFrame({'StartFrame': [15 * i for i in range(nsamples)], 'EndFrame': [15 * i + 15 for i in range(nsamples)]})
        df['Video'], df['Domain'] = 'vid', 'Gesture'
        ds.data = {'Custom_inference': df}
        g[name + "/nwindows"] = np.int64(len(ds))
        for w in range(len(ds)):
            videoname, snippets, flows, label, imp, dom = ds[w]
            assert videoname == 'vid' and int(label) == 0 and isinstance(snippets, tuple)
            for v in range(3):
                g[f"{name}/w{w}/rgb{v}"] = snippets[v][0, :, 0].numpy().astype(np.int64)
                g[f"{name}/w{w}/flow{v}"] = flows[v][0, :, 0].numpy().astype(np.int64)
            g[f"{name}/w{w}/imp_len"] = np.int64(imp.shape[1])
    np.savez_compressed(os.path.join(HERE, "sampler.npz"), **g)
    print("sampler.npz", len(g), "arrays")


if __name__ == "__main__":
    main()
