"""HDF5 feature files without h5py: sais_amd.hdf5_min against a fixture written by libhdf5 itself, and
its writer read back through libhdf5 where that library is installed (it is in the build image)."""
import os
import subprocess

import numpy as np
import pytest

import make_golden_h5 as G
from sais_amd import hdf5_min as H

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reps_libhdf5.h5")


def test_reader_on_a_libhdf5_written_file():
    got = H.read_h5(GOLDEN)
    want = G.expected_arrays()
    assert list(got.keys()) == sorted(want.keys())                     # f.keys() order = name order
    for k, a in want.items():
        assert got[k].dtype == np.float32 and got[k].shape == a.shape, k
        assert np.array_equal(got[k], a), k


@pytest.mark.parametrize("n", [0, 1, 3, 9, 40])
def test_writer_roundtrip(tmp_path, n):
    rng = np.random.default_rng(n)
    arrays = {"video_%03d" % (7 * i % 41): rng.standard_normal((1 + i % 4, 384)).astype(np.float32) for i in range(n)}
    if n >= 3:
        arrays["zero_rows"] = np.zeros((0, 384), np.float32)
    p = str(tmp_path / "w.h5")
    H.write_h5(p, arrays)
    got = H.read_h5(p)
    assert list(got.keys()) == sorted(arrays.keys())
    for k, a in arrays.items():
        assert np.array_equal(got[k], a) and got[k].shape == a.shape
    H.write_h5(p, {"only": np.ones((2, 384), np.float32)})             # mode 'w': earlier videos are gone (saveH5 :391)
    assert list(H.read_h5(p).keys()) == ["only"]


def test_writer_output_opens_in_libhdf5(tmp_path):
    L = G.load_libhdf5()
    if L is None:
        pytest.skip("libhdf5 not installed here")
    rng = np.random.default_rng(5)
    arrays = {"v%02d" % i: rng.standard_normal((2 + i, 384)).astype(np.float32) for i in range(11)}
    arrays["empty"] = np.zeros((0, 384), np.float32)
    arrays["f64_in"] = rng.standard_normal((3, 384))                    # written as f64, libhdf5 converts to f32 on read
    p = str(tmp_path / "w.h5")
    H.write_h5(p, arrays)
    for k, a in arrays.items():
        got = G.libhdf5_read(L, p, k)
        assert got.shape == a.shape, k
        assert np.array_equal(got, a.astype(np.float32)), k
    h5ls = "/opt/conda/bin/h5ls"
    if os.path.exists(h5ls):
        r = subprocess.run([h5ls, p], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.count("Dataset") == len(arrays), r.stdout + r.stderr


def test_rejects_what_it_does_not_implement(tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"not hdf5 at all" * 10)
    with pytest.raises(H.Hdf5FormatError):
        H.read_h5(str(p))
    with pytest.raises(ValueError):
        H.write_h5(str(p), {"a/b": np.zeros((1, 2), np.float32)})
