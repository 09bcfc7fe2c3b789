"""Minimal HDF5 reader / writer for the feature files of the SAIS path — no h5py, no libhdf5.

The reference stores per-frame ViT features as ONE flat group of 2-D float datasets, one per video label
(SAIS/scripts/extract_representations.py:389-407 `saveH5`: `h5py.File(path, 'w')`, `create_dataset(label, data=reps)`),
and reads them back with `h5py.File(path, 'r')` + `.get(videoname)` (prepare_dataset.py:1702-1703).  That subset of the
HDF5 file format (HDF5 File Format Specification v1/v2: superblock version 0, "old-style" groups = v1 B-tree + local
heap + symbol-table nodes, version-1 object headers, contiguous dataset layout) is what h5py writes by default and is
all this module implements:

  write_h5(path, {name: array})   -> a file h5py / h5dump / libhdf5 open: superblock v0, one root group, one contiguous
                                     little-endian IEEE-f32 (or f64 / i32 / i64) N-d dataset per key, truncating like mode 'w'
  read_h5(path) -> {name: ndarray} of every dataset directly under the root group (contiguous or compact layout,
                                     fixed-point / floating-point types of either endianness; chunked datasets raise)

Pinned by tests/test_hdf5.py against a fixture written by libhdf5 1.10.6 itself (tests/golden/reps_libhdf5.h5, generator
tests/golden/make_golden_h5.py) and, where libhdf5 is installed, by reading this writer's files back through it.
Host-side I/O only: nothing here touches the GPU.
"""
import os
import struct

import numpy as np

SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF
GROUP_INTERNAL_K = 16            # libhdf5 defaults (superblock fields)
GROUP_LEAF_K = 4


class Hdf5FormatError(ValueError):
    pass


# ------------------------------------------------------------------------------------------------- writer
def _pad8(b):
    return b + b"\x00" * (-len(b) % 8)


def _msg(mtype, data, flags=0):
    data = _pad8(data)
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


_NP_TYPES = {                    # dtype -> (datatype message body)
    "f4": struct.pack("<BBBBI", 0x11, 0x20, 31, 0, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127),
    "f8": struct.pack("<BBBBI", 0x11, 0x20, 63, 0, 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023),
    "i4": struct.pack("<BBBBI", 0x10, 0x08, 0, 0, 4) + struct.pack("<HH", 0, 32),
    "i8": struct.pack("<BBBBI", 0x10, 0x08, 0, 0, 8) + struct.pack("<HH", 0, 64),
}


def _dataset_header(arr, data_addr):
    """Version-1 object header of a contiguous dataset: dataspace, datatype, fill value, layout."""
    dims = arr.shape
    space = struct.pack("<BBB5x", 1, len(dims), 1) + b"".join(struct.pack("<Q", d) for d in dims) * 2   # dims, max dims
    dtype = _NP_TYPES[arr.dtype.str[1:]]
    fill = struct.pack("<BBBBI", 2, 2, 2, 1, 0)              # v2, alloc late, fill time ifset, defined, size 0 (default)
    layout = struct.pack("<BBQQ", 3, 1, data_addr if arr.nbytes else UNDEF, arr.nbytes)
    msgs = _msg(0x0001, space) + _msg(0x0003, dtype, 1) + _msg(0x0005, fill, 1) + _msg(0x0008, layout)
    return struct.pack("<BBHII4x", 1, 0, 4, 1, len(msgs)) + msgs


def write_h5(path, datasets):
    """Write {name: array} as a flat HDF5 file (truncating, like h5py mode 'w' in saveH5 :391)."""
    items = []
    for name, a in datasets.items():
        a = np.ascontiguousarray(a)
        if a.dtype.str[1:] not in _NP_TYPES:
            a = a.astype(np.float32)
        a = a.astype(a.dtype.newbyteorder("<"), copy=False)
        nb = name.encode("utf-8")
        if not nb or b"/" in nb or b"\x00" in nb:
            raise ValueError(f"invalid dataset name {name!r}")
        items.append((nb, a))
    items.sort(key=lambda t: t[0])                            # symbol-table entries are ordered by name (strcmp)
    n = len(items)
    leaf_k = max(GROUP_LEAF_K, (n + 1) // 2)                  # one symbol-table node holds 2 * leaf K entries
    if leaf_k > 0x7FFF:
        raise ValueError("too many datasets for one symbol-table node")

    # local heap data segment: "" at offset 0, then the names, then one free block
    heap = bytearray(8)
    name_off = []
    for nb, _ in items:
        name_off.append(len(heap))
        heap += _pad8(nb + b"\x00")
    free_off = len(heap)
    heap += struct.pack("<QQ", 1, 32) + b"\x00" * 16         # free block: next = 1 (none), size 32

    root_hdr_addr = 96
    root_msgs = 16 + 8 + 16                                   # header prefix + message header + symbol-table message
    btree_addr = root_hdr_addr + root_msgs
    btree_size = 24 + (2 * GROUP_INTERNAL_K + 1) * 8 + 2 * GROUP_INTERNAL_K * 8
    heap_addr = btree_addr + btree_size
    heap_data_addr = heap_addr + 32
    snod_addr = heap_data_addr + len(heap)
    snod_size = 8 + 2 * leaf_k * 40
    pos = snod_addr + snod_size
    hdr_addr, data_addr = [], []
    for nb, a in items:
        hdr_addr.append(pos)
        pos += len(_dataset_header(a, 0))
    pos = (pos + 63) // 64 * 64
    for nb, a in items:
        data_addr.append(pos)
        pos += (a.nbytes + 7) // 8 * 8
    eof = pos

    out = bytearray()
    out += SIG + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, leaf_k, GROUP_INTERNAL_K, 0)
    out += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    out += struct.pack("<QQI4xQQ", 0, root_hdr_addr, 1, btree_addr, heap_addr)      # root symbol-table entry
    assert len(out) == 96
    out += struct.pack("<BBHII4x", 1, 0, 1, 1, 24) + _msg(0x0011, struct.pack("<QQ", btree_addr, heap_addr))
    # B-tree leaf with one child (the symbol-table node); an empty group has zero entries
    bt = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if n else 0, UNDEF, UNDEF)
    bt += struct.pack("<QQQ", 0, snod_addr, name_off[-1] if n else 0)
    out += bt + b"\x00" * (btree_size - len(bt))
    out += b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), free_off, heap_data_addr) + heap
    sn = b"SNOD" + struct.pack("<BBH", 1, 0, n)
    for i in range(n):
        sn += struct.pack("<QQII16x", name_off[i], hdr_addr[i], 0, 0)
    out += sn + b"\x00" * (snod_size - len(sn))
    for i, (nb, a) in enumerate(items):
        assert len(out) == hdr_addr[i]
        out += _dataset_header(a, data_addr[i])
    with open(path, "wb") as fh:
        fh.write(out)
        for i, (nb, a) in enumerate(items):
            fh.write(b"\x00" * (data_addr[i] - fh.tell()))
            fh.write(a.tobytes())
        fh.write(b"\x00" * (eof - fh.tell()))
    return path


# ------------------------------------------------------------------------------------------------- reader
class _File:
    def __init__(self, path):
        # the file is MAPPED, not read: datasets come back as views of the mapping, so the resident footprint is the pages
        # actually touched (the reference opens its feature files lazily too: h5py .get per item, prepare_dataset.py:1702)
        import mmap
        with open(path, "rb") as fh:
            if os.fstat(fh.fileno()).st_size == 0:
                raise Hdf5FormatError(f"{path}: empty file")
            self.b = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
        b = self.b
        base = -1
        off = 0
        while off + 8 <= len(b):                              # the superblock may sit at 0, 512, 1024, ...
            if b[off:off + 8] == SIG:
                base = off
                break
            off = 512 if off == 0 else off * 2
        if base < 0:
            raise Hdf5FormatError(f"{path}: not an HDF5 file (signature not found)")
        ver = b[base + 8]
        if ver > 1:
            raise Hdf5FormatError(f"{path}: superblock version {ver} (libver='latest' files) is not supported; "
                                  "the reference writes version 0 (h5py default)")
        self.so, self.sl = b[base + 13], b[base + 14]
        if (self.so, self.sl) != (8, 8):
            raise Hdf5FormatError("only 8-byte offsets / lengths are supported")
        p = base + 24 + (4 if ver == 1 else 0)
        self.base_addr, _, self.eof, _ = struct.unpack_from("<QQQQ", b, p)
        self.base_addr += base
        p += 32
        _, self.root_hdr, cache, = struct.unpack_from("<QQI", b, p)
        self.root_scratch = struct.unpack_from("<QQ", b, p + 24) if cache == 1 else None

    def at(self, addr):
        return self.base_addr + addr

    # ---- object headers (version 1)
    def messages(self, addr):
        b = self.b
        p = self.at(addr)
        if b[p] != 1:
            raise Hdf5FormatError(f"object header version {b[p]} at {addr:#x} is not supported (expected 1)")
        nmsg, _, size = struct.unpack_from("<HII", b, p + 2)
        chunks = [(p + 16, size)]
        out = []
        while chunks and len(out) < nmsg:
            q, left = chunks.pop(0)
            end = q + left
            while q + 8 <= end and len(out) < nmsg:
                mtype, msize, flags = struct.unpack_from("<HHB", b, q)
                body = b[q + 8:q + 8 + msize]
                q += 8 + msize
                if flags & 2:
                    raise Hdf5FormatError("shared object-header messages are not supported")
                if mtype == 0x0010:                           # continuation
                    caddr, clen = struct.unpack_from("<QQ", body, 0)
                    chunks.append((self.at(caddr), clen))
                out.append((mtype, body))
        return out

    # ---- old-style group traversal
    def heap_data(self, heap_addr):
        p = self.at(heap_addr)
        if self.b[p:p + 4] != b"HEAP":
            raise Hdf5FormatError("local heap signature missing")
        size, _, daddr = struct.unpack_from("<QQQ", self.b, p + 8)
        d = self.at(daddr)
        return self.b[d:d + size]

    def snods(self, btree_addr):
        p = self.at(btree_addr)
        b = self.b
        if b[p:p + 4] != b"TREE":
            raise Hdf5FormatError("B-tree signature missing")
        ntype, level, used = struct.unpack_from("<BBH", b, p + 4)
        if ntype != 0:
            raise Hdf5FormatError("expected a group B-tree")
        q = p + 24
        for i in range(used):
            child, = struct.unpack_from("<Q", b, q + 8 + 16 * i)
            if level > 0:
                yield from self.snods(child)
            else:
                yield child

    def links(self):
        msgs = dict(self.messages(self.root_hdr))
        if 0x0011 in msgs:
            btree, heap = struct.unpack_from("<QQ", msgs[0x0011], 0)
        elif self.root_scratch:
            btree, heap = self.root_scratch
        else:
            raise Hdf5FormatError("root group has no symbol table (new-style groups are not supported)")
        names = self.heap_data(heap)
        b = self.b
        for sn in self.snods(btree):
            p = self.at(sn)
            if b[p:p + 4] != b"SNOD":
                raise Hdf5FormatError("symbol-table node signature missing")
            nsym, = struct.unpack_from("<H", b, p + 6)
            for i in range(nsym):
                noff, haddr = struct.unpack_from("<QQ", b, p + 8 + 40 * i)
                end = names.index(b"\x00", noff)
                yield names[noff:end].decode("utf-8"), haddr

    # ---- datasets
    @staticmethod
    def _dtype(body):
        cls, ver = body[0] & 0x0F, body[0] >> 4
        bits0 = body[1]
        size, = struct.unpack_from("<I", body, 4)
        order = ">" if bits0 & 1 else "<"
        if cls == 1:                                          # floating point: accept IEEE binary32 / binary64 only
            _, prec, eloc, esize, mloc, msize, bias = struct.unpack_from("<HHBBBBI", body, 8)
            if (size, prec, eloc, esize, mloc, msize, bias) == (4, 32, 23, 8, 0, 23, 127):
                return np.dtype(order + "f4")
            if (size, prec, eloc, esize, mloc, msize, bias) == (8, 64, 52, 11, 0, 52, 1023):
                return np.dtype(order + "f8")
            raise Hdf5FormatError("unsupported floating-point layout")
        if cls == 0:
            signed = bool(bits0 & 8)
            if size not in (1, 2, 4, 8):
                raise Hdf5FormatError("unsupported integer size")
            return np.dtype(order + ("i" if signed else "u") + str(size))
        raise Hdf5FormatError(f"unsupported datatype class {cls}")

    def _view(self, pos, dt, n, dims):
        """Contiguous dataset at file position `pos`: a read-only view of the mapping when the file's byte order is the
        host's (no copy), a converted copy otherwise."""
        if pos + n * dt.itemsize > len(self.b):
            raise Hdf5FormatError("dataset extends past the end of the file")
        a = np.frombuffer(self.b, dt, n, pos).reshape(dims)
        return a if dt.isnative else a.astype(dt.newbyteorder("="))

    def dataset(self, haddr):
        msgs = self.messages(haddr)
        by = {}
        for t, body in msgs:
            by.setdefault(t, body)
        if 0x0001 not in by or 0x0003 not in by or 0x0008 not in by:
            return None                                       # not a dataset (e.g. a sub-group)
        sp = by[0x0001]
        ver, rank = sp[0], sp[1]
        doff = 8 if ver == 1 else 4
        dims = struct.unpack_from("<%dQ" % rank, sp, doff) if rank else ()
        dt = self._dtype(by[0x0003])
        lay = by[0x0008]
        n = int(np.prod(dims)) if rank else 1
        if lay[0] == 3:
            if lay[1] == 1:
                addr, size = struct.unpack_from("<QQ", lay, 2)
                if addr != UNDEF:
                    return self._view(self.at(addr), dt, n, dims)
                raw = b""
            elif lay[1] == 0:
                size, = struct.unpack_from("<H", lay, 2)
                raw = lay[4:4 + size]
            else:
                raise Hdf5FormatError("chunked datasets are not supported (the reference writes contiguous ones)")
        elif lay[0] in (1, 2):
            rk, cls = lay[1], lay[2]
            if cls != 1:
                raise Hdf5FormatError("only contiguous version-1/2 layouts are supported")
            addr, = struct.unpack_from("<Q", lay, 8)
            return self._view(self.at(addr), dt, n, dims)
        else:
            raise Hdf5FormatError(f"layout message version {lay[0]} is not supported")
        if len(raw) < n * dt.itemsize:
            if len(raw) == 0:                                 # never written: libhdf5 returns the fill value (0)
                return np.zeros(dims, dt.newbyteorder("="))
            raise Hdf5FormatError("dataset extends past the end of the file")
        a = np.frombuffer(raw, dt, n).reshape(dims)
        return a.astype(dt.newbyteorder("="))


def read_h5(path):
    """{name: ndarray} of the datasets under the root group, in name order (= `for k in f.keys()`)."""
    f = _File(path)
    out = {}
    for name, haddr in f.links():
        a = f.dataset(haddr)
        if a is not None:
            out[name] = a
    return out
