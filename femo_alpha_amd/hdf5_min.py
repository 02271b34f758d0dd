"""A small read-only HDF5 parser for the datasets of dolfinx-written XDMF meshes (no h5py, no libhdf5).

The reference reads every mesh through ``dolfinx.io.XDMFFile(...).read_mesh(name="Grid")``
(femo_alpha/fea/utils_dolfinx.py:34-50): an XML file whose two ``DataItem`` entries point into an HDF5 file,
``<file>.h5:/Mesh/Grid/topology`` (int64, cells x vertices) and ``.../geometry`` (float64, points x 2|3).
h5py is not part of this image's python; the arrays are plain, so this module walks the file format itself
(HDF5 File Format Specification 3.0): superblock versions 0-3, version 1 and 2 object headers with continuation
blocks, old-style groups (symbol-table B-tree + local heap) and compact new-style groups (link messages),
dataspace / datatype / layout messages, contiguous, compact and unfiltered chunked storage (version 1 B-tree
index), little-endian fixed-point and IEEE floating-point element types.

Not supported, and reported as such: filters (compression), dense link storage (fractal heaps), version 4 chunk
indices, big-endian or compound types.  Use ``read_dataset(path, "/Mesh/Grid/topology")``.
"""
from __future__ import annotations

import struct

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class HDF5FormatError(ValueError):
    pass


class _File:
    def __init__(self, path):
        with open(path, "rb") as fh:
            self.buf = fh.read()
        self.path = path
        base = 0
        while self.buf[base:base + 8] != _SIG:                 # the superblock may sit at 0, 512, 1024, ...
            base = 512 if base == 0 else base * 2
            if base + 8 > len(self.buf):
                raise HDF5FormatError(f"{path}: not an HDF5 file (no superblock signature)")
        self.sb = base
        ver = self.buf[base + 8]
        if ver in (0, 1):
            self.O, self.L = self.buf[base + 13], self.buf[base + 14]
            p = base + 24 + (4 if ver == 1 else 0)
            self.base_addr = self._u(p, self.O)
            p += 4 * self.O                                        # base, free-space info, end of file, driver info
            # root group symbol table entry
            self.root_header = self._u(p + self.O, self.O)
            cache = self._u(p + 2 * self.O, 4)
            self.root_stab = None
            if cache == 1:
                sp = p + 2 * self.O + 8
                self.root_stab = (self._u(sp, self.O), self._u(sp + self.O, self.O))
        elif ver in (2, 3):
            self.O, self.L = self.buf[base + 9], self.buf[base + 10]
            p = base + 12
            self.base_addr = self._u(p, self.O)
            self.root_header = self._u(p + 3 * self.O, self.O)
            self.root_stab = None
        else:
            raise HDF5FormatError(f"{path}: unsupported superblock version {ver}")
        if self.O not in (4, 8) or self.L not in (4, 8):
            raise HDF5FormatError(f"{path}: unsupported offset/length sizes {self.O}/{self.L}")

    # ------------------------------------------------------------------ primitives
    def _u(self, pos, n):
        return int.from_bytes(self.buf[pos:pos + n], "little")

    def _addr(self, a):
        return self.base_addr + a

    # ------------------------------------------------------------------ object headers
    def messages(self, address):
        """[(type, flags, bytes)] of the object header at ``address`` (continuations followed)."""
        pos = self._addr(address)
        out = []
        if self.buf[pos:pos + 4] == b"OHDR":
            if self.buf[pos + 4] != 2:
                raise HDF5FormatError("unsupported object header version")
            flags = self.buf[pos + 5]
            p = pos + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            szlen = 1 << (flags & 3)
            size0 = self._u(p, szlen)
            p += szlen
            blocks = [(p, p + size0)]
            track = bool(flags & 0x04)
            while blocks:
                p, end = blocks.pop(0)
                while p + 4 <= end:
                    mtype, msize, mflags = self.buf[p], self._u(p + 1, 2), self.buf[p + 3]
                    p += 4 + (2 if track else 0)
                    data = self.buf[p:p + msize]
                    p += msize
                    if mtype == 0x10:
                        off, ln = self._u_from(data, 0, self.O), self._u_from(data, self.O, self.L)
                        cp = self._addr(off)
                        if self.buf[cp:cp + 4] != b"OCHK":
                            raise HDF5FormatError("bad object header continuation")
                        blocks.append((cp + 4, cp + ln - 4))       # signature in front, checksum behind
                    elif mtype != 0:
                        out.append((mtype, mflags, data))
            return out
        if self.buf[pos] != 1:
            raise HDF5FormatError(f"unsupported object header version {self.buf[pos]} at {address}")
        nmsg = self._u(pos + 2, 2)
        size = self._u(pos + 8, 4)
        blocks = [(pos + 16, pos + 16 + size)]
        while blocks and len(out) < nmsg + 64:
            p, end = blocks.pop(0)
            while p + 8 <= end:
                mtype, msize, mflags = self._u(p, 2), self._u(p + 2, 2), self.buf[p + 4]
                p += 8
                data = self.buf[p:p + msize]
                p += msize
                if mtype == 0x10:
                    blocks.append((self._addr(self._u_from(data, 0, self.O)),
                                   self._addr(self._u_from(data, 0, self.O)) + self._u_from(data, self.O, self.L)))
                elif mtype != 0:
                    out.append((mtype, mflags, data))
        return out

    @staticmethod
    def _u_from(data, pos, n):
        return int.from_bytes(data[pos:pos + n], "little")

    # ------------------------------------------------------------------ groups
    def _heap_name(self, heap_addr, offset):
        p = self._addr(heap_addr)
        if self.buf[p:p + 4] != b"HEAP":
            raise HDF5FormatError("bad local heap")
        data = self._addr(self._u(p + 8 + 2 * self.L, self.O))
        end = self.buf.index(b"\x00", data + offset)
        return self.buf[data + offset:end].decode()

    def _walk_group_btree(self, node_addr, heap_addr, out):
        p = self._addr(node_addr)
        if self.buf[p:p + 4] == b"SNOD":
            n = self._u(p + 6, 2)
            e = p + 8
            for _ in range(n):
                out[self._heap_name(heap_addr, self._u(e, self.O))] = self._u(e + self.O, self.O)
                e += 2 * self.O + 24
            return
        if self.buf[p:p + 4] != b"TREE" or self.buf[p + 4] != 0:
            raise HDF5FormatError("bad group B-tree node")
        used = self._u(p + 6, 2)
        e = p + 8 + 2 * self.O
        for _ in range(used):
            e += self.L                                           # key
            self._walk_group_btree(self._u(e, self.O), heap_addr, out)
            e += self.O

    def children(self, header_addr, stab=None):
        """{name: object header address} of a group."""
        out = {}
        for mtype, _, d in self.messages(header_addr):
            if mtype == 0x11:                                     # symbol table message (old-style group)
                stab = (self._u_from(d, 0, self.O), self._u_from(d, self.O, self.O))
            elif mtype == 0x06:                                   # link message (compact new-style group)
                if d[0] != 1:
                    raise HDF5FormatError("unsupported link message version")
                fl = d[1]
                p = 2
                ltype = 0
                if fl & 0x08:
                    ltype = d[p]; p += 1
                if fl & 0x04:
                    p += 8
                if fl & 0x10:
                    p += 1
                nlen_sz = 1 << (fl & 3)
                nlen = self._u_from(d, p, nlen_sz); p += nlen_sz
                name = d[p:p + nlen].decode(); p += nlen
                if ltype == 0:
                    out[name] = self._u_from(d, p, self.O)
            elif mtype == 0x02:                                   # link info: dense storage if a fractal heap is named
                if self._u_from(d, 2 + (8 if d[1] & 1 else 0), self.O) != (_UNDEF >> (8 * (8 - self.O))):
                    raise HDF5FormatError("groups with dense link storage (fractal heap) are not supported by hdf5_min")
        if stab is not None:
            self._walk_group_btree(stab[0], stab[1], out)
        return out

    def resolve(self, path):
        addr, stab = self.root_header, self.root_stab
        for part in [s for s in path.split("/") if s]:
            kids = self.children(addr, stab)
            if part not in kids:
                raise KeyError(f"{self.path}: no object '{part}' on the way to '{path}' (have {sorted(kids)})")
            addr, stab = kids[part], None
        return addr

    # ------------------------------------------------------------------ datasets
    def dataset(self, path):
        shape = dtype = layout = None
        for mtype, _, d in self.messages(self.resolve(path)):
            if mtype == 0x01:
                ver, rank, fl = d[0], d[1], d[2]
                p = 8 if ver == 1 else 4
                shape = tuple(self._u_from(d, p + i * self.L, self.L) for i in range(rank))
            elif mtype == 0x03:
                cls, bits0, size = d[0] & 0x0F, d[1], self._u_from(d, 4, 4)
                if bits0 & 1:
                    raise HDF5FormatError("big-endian data are not supported by hdf5_min")
                if cls == 0:
                    dtype = np.dtype(("<i" if bits0 & 0x08 else "<u") + str(size))
                elif cls == 1 and size in (4, 8):
                    dtype = np.dtype("<f" + str(size))
                else:
                    raise HDF5FormatError(f"unsupported datatype class {cls} (size {size})")
            elif mtype == 0x08:
                layout = d
            elif mtype == 0x0B:
                raise HDF5FormatError("filtered (compressed) datasets are not supported by hdf5_min")
        if shape is None or dtype is None or layout is None:
            raise HDF5FormatError(f"'{path}' is not a simple dataset")
        n = int(np.prod(shape)) if shape else 1
        ver = layout[0]
        if ver not in (3, 4):
            raise HDF5FormatError(f"unsupported data layout version {ver}")
        cls = layout[1]
        if cls == 0:                                              # compact
            size = self._u_from(layout, 2, 2)
            raw = layout[4:4 + size]
            return np.frombuffer(raw, dtype=dtype, count=n).reshape(shape).copy()
        if cls == 1:                                              # contiguous
            addr = self._u_from(layout, 2, self.O)
            if addr == (_UNDEF >> (8 * (8 - self.O))):
                return np.zeros(shape, dtype=dtype)               # never written
            return np.frombuffer(self.buf, dtype=dtype, count=n, offset=self._addr(addr)).reshape(shape).copy()
        if cls == 2 and ver == 3:                                 # chunked, version 1 B-tree index, no filters
            ndim = layout[2]
            btree = self._u_from(layout, 3, self.O)
            cdims = tuple(self._u_from(layout, 3 + self.O + 4 * i, 4) for i in range(ndim - 1))
            out = np.zeros(shape, dtype=dtype)
            self._read_chunks(btree, ndim, cdims, out)
            return out
        raise HDF5FormatError(f"unsupported data layout class {cls} (version {ver})")

    def _read_chunks(self, node_addr, ndim, cdims, out):
        p = self._addr(node_addr)
        if self.buf[p:p + 4] != b"TREE" or self.buf[p + 4] != 1:
            raise HDF5FormatError("bad chunk B-tree node")
        level, used = self.buf[p + 5], self._u(p + 6, 2)
        e = p + 8 + 2 * self.O
        keylen = 8 + 8 * ndim
        for _ in range(used):
            nbytes, mask = self._u(e, 4), self._u(e + 4, 4)
            offs = tuple(self._u(e + 8 + 8 * i, 8) for i in range(ndim - 1))
            child = self._u(e + keylen, self.O)
            e += keylen + self.O
            if level > 0:
                self._read_chunks(child, ndim, cdims, out)
                continue
            if mask:
                raise HDF5FormatError("filtered chunks are not supported by hdf5_min")
            chunk = np.frombuffer(self.buf, dtype=out.dtype, count=int(np.prod(cdims)), offset=self._addr(child)).reshape(cdims)
            sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, out.shape))
            out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]


def read_dataset(path, name):
    """numpy array of dataset ``name`` (e.g. "/Mesh/Grid/geometry") in the HDF5 file ``path``."""
    return _File(path).dataset(name)


def list_group(path, name="/"):
    """Names below group ``name``."""
    f = _File(path)
    addr = f.resolve(name)
    return sorted(f.children(addr, f.root_stab if addr == f.root_header else None))
