#!/usr/bin/env python3
"""Writes tests/golden/reps_libhdf5.h5 with libhdf5 ITSELF (1.10.6, /opt/conda/lib/libhdf5.so in the build container,
driven through ctypes: H5Fcreate / H5Screate_simple / H5Dcreate2 / H5Dwrite — the calls h5py makes for the reference's
saveH5, extract_representations.py:389-407).  The file pins sais_amd.hdf5_min.read_h5: 12 datasets (so libhdf5 splits
the root group over several symbol-table nodes), one of them with zero rows.  Contents are seeded:
dataset i = default_rng(i).standard_normal((rows_i, 384)).astype(float32), see EXPECTED below.

    python tests/golden/make_golden_h5.py            (needs libhdf5; not needed to run the tests)
"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
EXPECTED = [("vidA", 5, 0), ("video_b", 3, 1), ("empty", 0, 2)] + [("clip_%02d" % i, 1 + i % 3, 10 + i) for i in range(9)]


def expected_arrays():
    return {name: np.random.default_rng(seed).standard_normal((rows, 384)).astype(np.float32)
            for name, rows, seed in EXPECTED}


def load_libhdf5():
    for cand in (os.environ.get("LIBHDF5"), "/opt/conda/lib/libhdf5.so", "libhdf5.so", "libhdf5_serial.so"):
        if not cand:
            continue
        try:
            L = ctypes.CDLL(cand)
        except OSError:
            continue
        hid = ctypes.c_int64
        L.H5open()
        L.H5Fcreate.restype = hid; L.H5Fcreate.argtypes = [ctypes.c_char_p, ctypes.c_uint, hid, hid]
        L.H5Fopen.restype = hid; L.H5Fopen.argtypes = [ctypes.c_char_p, ctypes.c_uint, hid]
        L.H5Screate_simple.restype = hid; L.H5Screate_simple.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.H5Dcreate2.restype = hid; L.H5Dcreate2.argtypes = [hid, ctypes.c_char_p, hid, hid, hid, hid, hid]
        L.H5Dopen2.restype = hid; L.H5Dopen2.argtypes = [hid, ctypes.c_char_p, hid]
        L.H5Dget_space.restype = hid; L.H5Dget_space.argtypes = [hid]
        L.H5Sget_simple_extent_ndims.argtypes = [hid]
        L.H5Sget_simple_extent_dims.argtypes = [hid, ctypes.c_void_p, ctypes.c_void_p]
        L.H5Dwrite.argtypes = [hid, hid, hid, hid, hid, ctypes.c_void_p]
        L.H5Dread.argtypes = [hid, hid, hid, hid, hid, ctypes.c_void_p]
        for f in ("H5Dclose", "H5Sclose", "H5Fclose"):
            getattr(L, f).argtypes = [hid]
        L.f32 = hid.in_dll(L, "H5T_NATIVE_FLOAT_g").value
        return L
    return None


def libhdf5_write(L, path, arrays):
    fid = L.H5Fcreate(path.encode(), 2, 0, 0)                 # H5F_ACC_TRUNC
    assert fid >= 0
    for name, a in arrays.items():
        dims = (ctypes.c_uint64 * a.ndim)(*a.shape)
        sid = L.H5Screate_simple(a.ndim, dims, None)
        did = L.H5Dcreate2(fid, name.encode(), L.f32, sid, 0, 0, 0)
        assert did >= 0
        if a.size:
            assert L.H5Dwrite(did, L.f32, 0, 0, 0, a.ctypes.data) >= 0
        L.H5Dclose(did); L.H5Sclose(sid)
    L.H5Fclose(fid)


def libhdf5_read(L, path, name):
    fid = L.H5Fopen(path.encode(), 0, 0)                      # H5F_ACC_RDONLY
    assert fid >= 0, "libhdf5 cannot open " + path
    did = L.H5Dopen2(fid, name.encode(), 0)
    assert did >= 0, "libhdf5 cannot open dataset " + name
    sid = L.H5Dget_space(did)
    nd = L.H5Sget_simple_extent_ndims(sid)
    dims = (ctypes.c_uint64 * nd)()
    L.H5Sget_simple_extent_dims(sid, dims, None)
    a = np.empty(tuple(dims), np.float32)
    if a.size:
        assert L.H5Dread(did, L.f32, 0, 0, 0, a.ctypes.data) >= 0
    L.H5Dclose(did); L.H5Sclose(sid); L.H5Fclose(fid)
    return a


if __name__ == "__main__":
    L = load_libhdf5()
    if L is None:
        sys.exit("libhdf5 not found")
    out = os.path.join(HERE, "reps_libhdf5.h5")
    libhdf5_write(L, out, expected_arrays())
    print("wrote", out, os.path.getsize(out), "bytes")
