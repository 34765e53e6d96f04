"""Minimal HDF5 reader (pure Python + numpy) for the reference's capture files.

The reference reads its captures with h5py (captured_data.py:94-108, 136-149: datasets ``cam_proj``, ``cam_k``,
``screen_position``, ``mask``, ``ray_origin``, ``ray_dir`` at the root of ``<name>.h5``); h5py is not part of this
image.  This module reads exactly that kind of file -- numeric N-d datasets in (nested) groups:

    with hdf5_lite.File(path) as f:        # also usable without ``with``
        f.keys(); "mask" in f
        a = f["cam_proj"][...]             # numpy array;  f["mask"][i] reads one item along axis 0
        f["screen_position"].shape, .dtype

Supported (HDF5 File Format Specification, versions 1-3 of the superblock): superblock 0/1/2/3; old-style groups
(symbol table: v1 B-tree + local heap) and new-style groups with compact link messages; object headers v1 and v2 with
continuation blocks; fixed-point and IEEE floating-point datatypes of either byte order; simple dataspaces; compact,
contiguous and chunked (v1 B-tree) layouts with the deflate, shuffle and fletcher32 filters; fill with zeros for
unallocated storage.  Anything else (dense link storage, layout version 4 chunk indexes, compound / variable-length
types, other filters) raises ``Hdf5Unsupported`` with the feature named -- convert such a file with tools/h5_to_npz.py
where h5py exists.  Validated in tests/test_hdf5_lite.py against files written by the HDF5 library itself and by h5py.
"""
from __future__ import annotations

import mmap
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5Unsupported(NotImplementedError):
    pass


class _Reader:
    def __init__(self, buf):
        self.b = buf
        self.osz = 8     # size of offsets
        self.lsz = 8     # size of lengths

    def u(self, pos, n):
        return int.from_bytes(self.b[pos:pos + n], "little")

    def off(self, pos):
        v = self.u(pos, self.osz)
        return UNDEF if v == (1 << (8 * self.osz)) - 1 else v

    def length(self, pos):
        return self.u(pos, self.lsz)


def _pad8(n):
    return (n + 7) & ~7


class Dataset:
    def __init__(self, f, name, msgs):
        self._f, self.name = f, name
        space = msgs.get(0x0001)
        dtype = msgs.get(0x0003)
        layout = msgs.get(0x0008)
        if space is None or dtype is None or layout is None:
            raise Hdf5Unsupported(f"{name}: not a dataset (dataspace / datatype / layout message missing)")
        self.shape = self._parse_space(space)
        self.dtype = self._parse_dtype(dtype)
        self._filters = self._parse_filters(msgs.get(0x000B))
        self._parse_layout(layout)

    # ---- messages
    def _parse_space(self, m):
        ver, rank, flags = m[0], m[1], m[2]
        if ver == 1:
            p = 8
        elif ver == 2:
            if m[3] == 2:
                raise Hdf5Unsupported(f"{self.name}: null dataspace")
            p = 4
        else:
            raise Hdf5Unsupported(f"{self.name}: dataspace message version {ver}")
        L = self._f._r.lsz
        return tuple(int.from_bytes(m[p + L * k:p + L * (k + 1)], "little") for k in range(rank))

    def _parse_dtype(self, m):
        cls, ver = m[0] & 0x0F, m[0] >> 4
        bits0 = m[1]
        size = int.from_bytes(m[4:8], "little")
        order = ">" if bits0 & 1 else "<"
        if cls == 0:                                        # fixed point
            kind = "i" if bits0 & 0x08 else "u"
            if size not in (1, 2, 4, 8):
                raise Hdf5Unsupported(f"{self.name}: {size}-byte integer")
            return np.dtype(f"{order}{kind}{size}")
        if cls == 1:                                        # floating point (IEEE layouts only)
            if bits0 & 0x40:
                raise Hdf5Unsupported(f"{self.name}: VAX byte order")
            if size not in (2, 4, 8):
                raise Hdf5Unsupported(f"{self.name}: {size}-byte float")
            return np.dtype(f"{order}f{size}")
        names = {2: "time", 3: "string", 4: "bitfield", 5: "opaque", 6: "compound", 7: "reference", 8: "enum", 9: "variable-length", 10: "array"}
        raise Hdf5Unsupported(f"{self.name}: datatype class {names.get(cls, cls)} (version {ver})")

    def _parse_filters(self, m):
        if m is None:
            return []
        ver, n = m[0], m[1]
        p = 8 if ver == 1 else 2
        out = []
        for _ in range(n):
            fid = int.from_bytes(m[p:p + 2], "little"); p += 2
            name_len = 0
            if ver == 1 or fid >= 256:
                name_len = int.from_bytes(m[p:p + 2], "little"); p += 2
            p += 2                                           # flags
            ncd = int.from_bytes(m[p:p + 2], "little"); p += 2
            p += _pad8(name_len) if ver == 1 else name_len
            cd = [int.from_bytes(m[p + 4 * k:p + 4 * k + 4], "little") for k in range(ncd)]
            p += 4 * ncd
            if ver == 1 and ncd % 2:
                p += 4
            if fid not in (1, 2, 3):
                raise Hdf5Unsupported(f"{self.name}: filter id {fid} (only deflate, shuffle, fletcher32)")
            out.append((fid, cd))
        return out

    def _parse_layout(self, m):
        r = self._f._r
        ver = m[0]
        self._chunk = None
        if ver == 3:
            cls = m[1]
            if cls == 0:
                size = int.from_bytes(m[2:4], "little")
                self._kind, self._data = "compact", bytes(m[4:4 + size])
            elif cls == 1:
                self._kind = "contiguous"
                self._addr = int.from_bytes(m[2:2 + r.osz], "little")
                if self._addr == (1 << (8 * r.osz)) - 1:
                    self._addr = UNDEF
            elif cls == 2:
                dim = m[2]
                self._kind = "chunked"
                self._btree = int.from_bytes(m[3:3 + r.osz], "little")
                if self._btree == (1 << (8 * r.osz)) - 1:
                    self._btree = UNDEF
                p = 3 + r.osz
                dims = [int.from_bytes(m[p + 4 * k:p + 4 * k + 4], "little") for k in range(dim)]
                self._chunk = tuple(dims[:-1])
            else:
                raise Hdf5Unsupported(f"{self.name}: layout class {cls}")
        elif ver in (1, 2):
            dim, cls = m[1], m[2]
            p = 8
            addr = UNDEF
            if cls != 0:
                addr = int.from_bytes(m[p:p + r.osz], "little"); p += r.osz
            dims = [int.from_bytes(m[p + 4 * k:p + 4 * k + 4], "little") for k in range(dim)]
            p += 4 * dim
            if cls == 0:
                size = int.from_bytes(m[p:p + 4], "little")
                self._kind, self._data = "compact", bytes(m[p + 4:p + 4 + size])
            elif cls == 1:
                self._kind, self._addr = "contiguous", addr
            else:
                self._kind, self._btree, self._chunk = "chunked", addr, tuple(dims[:-1])
        else:
            raise Hdf5Unsupported(f"{self.name}: data layout message version {ver} (written with libver='latest'?)")

    # ---- data
    def _unfilter(self, raw, mask):
        for k in range(len(self._filters) - 1, -1, -1):          # reverse order of the pipeline
            fid, _ = self._filters[k]
            if mask & (1 << k):
                continue
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 3:
                raw = raw[:-4]
            elif fid == 2:
                es = self.dtype.itemsize
                n = len(raw) // es
                raw = np.frombuffer(raw[:n * es], dtype=np.uint8).reshape(es, n).T.tobytes() + raw[n * es:]
        return raw

    def _read_chunked(self, lo=0, hi=None):
        """Rows [lo, hi) along axis 0 (default: everything).  Only the chunks that overlap the slab are decompressed,
        so indexing a [72, P, 3] dataset view by view costs one view's chunks per index, not the whole dataset."""
        hi = (self.shape[0] if self.shape else 1) if hi is None else hi
        out = np.zeros((hi - lo,) + tuple(self.shape[1:]), dtype=self.dtype) if self.shape else np.zeros((), dtype=self.dtype)
        if self._btree == UNDEF or out.size == 0:
            return out
        r = self._f._r
        rank = len(self.shape)
        base = self._f._base

        def walk(addr):
            p = base + addr
            if r.b[p:p + 4] != b"TREE" or r.b[p + 4] != 1:
                raise Hdf5Unsupported(f"{self.name}: bad chunk B-tree node")
            level, used = r.b[p + 5], r.u(p + 6, 2)
            p += 8 + 2 * r.osz
            key = 8 + 8 * (rank + 1)
            for _ in range(used):
                size, mask = r.u(p, 4), r.u(p + 4, 4)
                offs = [r.u(p + 8 + 8 * k, 8) for k in range(rank)]
                child = r.off(p + key)
                p += key + r.osz
                if level > 0:
                    walk(child)
                    continue
                if rank and (offs[0] >= hi or offs[0] + self._chunk[0] <= lo):
                    continue                                  # chunk outside the requested rows: not even decompressed
                raw = self._unfilter(bytes(r.b[base + child:base + child + size]), mask)
                chunk = np.frombuffer(raw, dtype=self.dtype, count=int(np.prod(self._chunk))).reshape(self._chunk)
                if not rank:
                    out[()] = chunk.reshape(())
                    continue
                a0, b0 = max(offs[0], lo), min(offs[0] + self._chunk[0], hi, self.shape[0])
                sel_out = (slice(a0 - lo, b0 - lo),) + tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs[1:], self._chunk[1:], self.shape[1:]))
                sel_in = (slice(a0 - offs[0], b0 - offs[0]),) + tuple(slice(0, s.stop - s.start) for s in sel_out[1:])
                out[sel_out] = chunk[sel_in]

        walk(self._btree)
        return out

    def read(self):
        n = int(np.prod(self.shape)) if self.shape else 1
        if self._kind == "compact":
            return np.frombuffer(self._data, dtype=self.dtype, count=n).reshape(self.shape).copy()
        if self._kind == "contiguous":
            if self._addr == UNDEF:
                return np.zeros(self.shape, dtype=self.dtype)
            a = self._f._base + self._addr
            return np.frombuffer(self._f._r.b, dtype=self.dtype, count=n, offset=a).reshape(self.shape).copy()
        return self._read_chunked()

    def __getitem__(self, key):
        if self._kind == "contiguous" and self._addr != UNDEF and isinstance(key, (int, np.integer)) and len(self.shape) >= 1:
            i = int(key) + (self.shape[0] if key < 0 else 0)          # one item along axis 0 without touching the rest
            if not 0 <= i < self.shape[0]:
                raise IndexError(key)
            inner = self.shape[1:]
            n = int(np.prod(inner)) if inner else 1
            a = self._f._base + self._addr + i * n * self.dtype.itemsize
            return np.frombuffer(self._f._r.b, dtype=self.dtype, count=n, offset=a).reshape(inner).copy()
        if self._kind == "chunked" and len(self.shape) >= 1:
            # integer or contiguous slice along axis 0: decode only the overlapping chunks
            if isinstance(key, (int, np.integer)):
                i = int(key) + (self.shape[0] if key < 0 else 0)
                if not 0 <= i < self.shape[0]:
                    raise IndexError(key)
                return self._read_chunked(i, i + 1)[0]
            if isinstance(key, slice) and key.step in (None, 1):
                lo, hi, _ = key.indices(self.shape[0])
                return self._read_chunked(lo, max(lo, hi))
        return self.read()[key]

    def __len__(self):
        return self.shape[0]

    def __array__(self, dtype=None):
        a = self.read()
        return a if dtype is None else a.astype(dtype)


class Group:
    def __init__(self, f, name, links):
        self._f, self.name, self._links = f, name, links

    def keys(self):
        return list(self._links)

    def __iter__(self):
        return iter(self._links)

    def __contains__(self, k):
        return k.strip("/").split("/")[0] in self._links if "/" in k.strip("/") else k.strip("/") in self._links

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split("/") if p]:
            if not isinstance(node, Group) or part not in node._links:
                raise KeyError(path)
            node = node._f._open(node._links[part], (node.name.rstrip("/") + "/" + part))
        return node


class File(Group):
    def __init__(self, path):
        self._fh = open(path, "rb")
        try:
            buf = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)     # captures are several GB: map, do not read
        except ValueError:                                                    # empty file
            self._fh.close()
            raise ValueError(f"{path}: not an HDF5 file")
        self._map = buf
        self._r = r = _Reader(memoryview(buf))
        sig = b"\x89HDF\r\n\x1a\n"
        base = 0
        while buf[base:base + 8] != sig:                     # the superblock may sit at 0, 512, 1024, ...
            base = 512 if base == 0 else base * 2
            if base >= len(buf):
                self.close()
                raise ValueError(f"{path}: not an HDF5 file")
        ver = buf[base + 8]
        if ver in (0, 1):
            r.osz, r.lsz = buf[base + 13], buf[base + 14]
            p = base + 24 + (4 if ver == 1 else 0)
            self._base = r.off(p)
            p += 4 * r.osz                                   # base, free-space, end-of-file, driver-info addresses
            root_header = r.off(p + r.osz)                   # root symbol-table entry: link name offset, object header address
        elif ver in (2, 3):
            r.osz, r.lsz = buf[base + 9], buf[base + 10]
            p = base + 12
            self._base = r.off(p)
            root_header = r.off(p + 3 * r.osz)
        else:
            raise Hdf5Unsupported(f"{path}: superblock version {ver}")
        if self._base == UNDEF:
            self._base = 0
        self.filename = path
        root = self._open(root_header, "/")
        if not isinstance(root, Group):
            raise Hdf5Unsupported(f"{path}: root object is not a group")
        Group.__init__(self, self, "/", root._links)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def close(self):
        """Arrays already read are copies; datasets of a closed file can no longer be read."""
        r, self._r = getattr(self, "_r", None), None
        if r is not None:
            r.b.release()
            self._map.close()
            self._fh.close()

    # ---- object headers
    def _messages(self, addr):
        """{type: last message body} plus the list of link messages, for the object header at ``addr``."""
        r, base = self._r, self._base
        p = base + addr
        msgs, links = {}, []

        def take(mtype, body):
            if mtype == 0x0006:
                links.append(body)
            else:
                msgs[mtype] = body

        if r.b[p:p + 4] == b"OHDR":                           # version 2
            flags = r.b[p + 5]
            q = p + 6
            if flags & 0x20:
                q += 16
            if flags & 0x10:
                q += 4
            csz = 1 << (flags & 3)
            chunk0 = r.u(q, csz); q += csz
            blocks = [(q, q + chunk0)]
            track = bool(flags & 0x04)
            while blocks:
                q, end = blocks.pop(0)
                while q + 4 <= end:
                    mtype, size = r.b[q], r.u(q + 1, 2)
                    q += 4 + (2 if track else 0)
                    body = r.b[q:q + size]
                    if mtype == 0x10:
                        a, ln = r.off(q), r.length(q + r.osz)
                        if r.b[base + a:base + a + 4] != b"OCHK":
                            raise Hdf5Unsupported("bad object header continuation block")
                        blocks.append((base + a + 4, base + a + ln - 4))
                    elif mtype != 0:
                        take(mtype, body)
                    q += size
            return msgs, links
        if r.b[p] != 1:
            raise Hdf5Unsupported(f"object header version {r.b[p]} at {addr}")
        n_msgs, size = r.u(p + 2, 2), r.u(p + 8, 4)
        blocks = [(p + 16, p + 16 + size)]
        seen = 0
        while blocks and seen < n_msgs:
            q, end = blocks.pop(0)
            while q + 8 <= end and seen < n_msgs:
                mtype, msize = r.u(q, 2), r.u(q + 2, 2)
                body = r.b[q + 8:q + 8 + msize]
                seen += 1
                if mtype == 0x10:
                    a, ln = r.off(q + 8), r.length(q + 8 + r.osz)
                    blocks.append((base + a, base + a + ln))
                elif mtype != 0:
                    take(mtype, body)
                q += 8 + msize
        return msgs, links

    def _symbol_table(self, btree, heap):
        r, base = self._r, self._base
        h = base + heap
        if r.b[h:h + 4] != b"HEAP":
            raise Hdf5Unsupported("bad local heap")
        data = base + r.off(h + 8 + 2 * r.lsz)
        out = {}

        def name_at(o):
            e = data + o
            end = e
            while r.b[end] != 0:
                end += 1
            return bytes(r.b[e:end]).decode("utf-8")

        def walk(addr):
            p = base + addr
            if r.b[p:p + 4] == b"SNOD":
                n = r.u(p + 6, 2)
                q = p + 8
                for _ in range(n):
                    out[name_at(r.off(q))] = r.off(q + r.osz)
                    q += 2 * r.osz + 24
                return
            if r.b[p:p + 4] != b"TREE" or r.b[p + 4] != 0:
                raise Hdf5Unsupported("bad group B-tree node")
            used = r.u(p + 6, 2)
            q = p + 8 + 2 * r.osz + r.lsz                     # skip key 0
            for _ in range(used):
                walk(r.off(q))
                q += r.osz + r.lsz

        if btree != UNDEF:
            walk(btree)
        return out

    def _link_message(self, m):
        r = self._r
        ver, flags = m[0], m[1]
        p = 2
        ltype = 0
        if flags & 0x08:
            ltype = m[p]; p += 1
        if flags & 0x04:
            p += 8
        if flags & 0x10:
            p += 1
        nsz = 1 << (flags & 3)
        nlen = int.from_bytes(m[p:p + nsz], "little"); p += nsz
        name = bytes(m[p:p + nlen]).decode("utf-8"); p += nlen
        if ltype != 0:
            raise Hdf5Unsupported(f"soft / external link {name!r}")
        return name, int.from_bytes(m[p:p + r.osz], "little")

    def _open(self, addr, name):
        if addr == UNDEF:
            raise Hdf5Unsupported(f"{name}: soft link (no object header)")
        msgs, links = self._messages(addr)
        if 0x0011 in msgs:                                     # old-style group
            m = msgs[0x0011]
            r = self._r
            return Group(self, name, self._symbol_table(int.from_bytes(m[:r.osz], "little"), int.from_bytes(m[r.osz:2 * r.osz], "little")))
        if links or 0x0002 in msgs:                            # new-style group
            if 0x0002 in msgs:
                m = msgs[0x0002]
                p = 2 + (8 if m[1] & 1 else 0)
                fractal = int.from_bytes(m[p:p + self._r.osz], "little")
                if fractal != (1 << (8 * self._r.osz)) - 1:
                    raise Hdf5Unsupported(f"{name}: group with dense link storage (fractal heap)")
            return Group(self, name, dict(self._link_message(m) for m in links))
        return Dataset(self, name, msgs)


# ---------------------------------------------------------------------------------------------------------------
# writer: the plainest file the format allows (what h5py's default ``create_dataset`` of a numeric array produces in
# its ``libver='earliest'`` form): superblock 0, one old-style root group (symbol table = v1 B-tree + local heap + one
# symbol node), version-1 object headers, contiguous little-endian datasets.  Used by tools/make_capture.py to write
# capture files with the reference's schema (captured_data.py:94-108) where h5py is not installed.
# ---------------------------------------------------------------------------------------------------------------
def _dtype_message(dt):
    dt = np.dtype(dt)
    if dt.kind == "f" and dt.itemsize in (4, 8):
        exp_loc, exp_size, man_size, bias = (23, 8, 23, 127) if dt.itemsize == 4 else (52, 11, 52, 1023)
        head = bytes([0x11, 0x20, 8 * dt.itemsize - 1, 0]) + dt.itemsize.to_bytes(4, "little")
        return head + (0).to_bytes(2, "little") + (8 * dt.itemsize).to_bytes(2, "little") + bytes([exp_loc, exp_size, 0, man_size]) + bias.to_bytes(4, "little")
    if dt.kind in "iub" and dt.itemsize in (1, 2, 4, 8):
        signed = 0x08 if dt.kind == "i" else 0
        head = bytes([0x10, signed, 0, 0]) + dt.itemsize.to_bytes(4, "little")
        return head + (0).to_bytes(2, "little") + (8 * dt.itemsize).to_bytes(2, "little")
    raise Hdf5Unsupported(f"cannot write dtype {dt}")


def _v1_message(mtype, body):
    body = body + bytes(_pad8(len(body)) - len(body))
    return mtype.to_bytes(2, "little") + len(body).to_bytes(2, "little") + bytes(4) + body


def _v1_header(messages):
    body = b"".join(messages)
    return bytes([1, 0]) + len(messages).to_bytes(2, "little") + (1).to_bytes(4, "little") + len(body).to_bytes(4, "little") + bytes(4) + body


def write_simple(path, arrays):
    """Write ``{name: array}`` as contiguous datasets at the root of a new HDF5 file (at most 8 datasets: one symbol node)."""
    names = sorted(arrays)                                # symbol-node entries are ordered by name
    if not 1 <= len(names) <= 8:
        raise ValueError("write_simple takes 1..8 datasets")
    u8 = lambda v: int(v).to_bytes(8, "little")
    arrs = {k: np.ascontiguousarray(arrays[k]) for k in names}
    for k in names:
        if arrs[k].dtype == np.bool_:
            arrs[k] = arrs[k].astype(np.uint8)
        arrs[k] = arrs[k].astype(arrs[k].dtype.newbyteorder("<"), copy=False)
    # local heap data: "" at offset 0, then the names, each NUL-terminated and 8-byte aligned
    heap_data, name_off = bytearray(8), {}
    for k in names:
        name_off[k] = len(heap_data)
        raw = k.encode("utf-8") + b"\0"
        heap_data += raw + bytes(_pad8(len(raw)) - len(raw))
    # layout of the file
    pos = 96                                              # superblock (56 bytes) + root symbol-table entry (40)
    root_hdr = pos
    root_hdr_len = 16 + 8 + 16                            # prefix + one symbol-table message
    pos = _pad8(pos + root_hdr_len)
    btree = pos; pos += 24 + 33 * 8 + 32 * 8              # full-size node for internal K = 16
    snod = pos; pos += 8 + 8 * 40                         # 2 x leaf K = 8 entries
    heap = pos; pos += 32
    heap_seg = pos; pos = _pad8(pos + len(heap_data))
    hdr_at, data_at = {}, {}
    hdrs = {}
    for k in names:
        a = arrs[k]
        space = bytes([1, a.ndim, 0, 0, 0, 0, 0, 0]) + b"".join(u8(s) for s in a.shape)
        hdr_at[k] = pos
        hdrs[k] = (space, _dtype_message(a.dtype))
        pos = _pad8(pos + 16 + (8 + _pad8(len(space))) + (8 + _pad8(len(hdrs[k][1]))) + (8 + 24))
    for k in names:
        data_at[k] = pos
        pos = _pad8(pos + arrs[k].nbytes)
    eof = pos
    with open(path, "wb") as f:
        sb = b"\x89HDF\r\n\x1a\n" + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + (4).to_bytes(2, "little") + (16).to_bytes(2, "little") + bytes(4)
        sb += u8(0) + u8(UNDEF) + u8(eof) + u8(UNDEF)
        sb += u8(0) + u8(root_hdr) + (1).to_bytes(4, "little") + bytes(4) + u8(btree) + u8(heap)     # root symbol-table entry (cached)
        assert len(sb) == 96
        f.write(sb)
        f.write(_v1_header([_v1_message(0x0011, u8(btree) + u8(heap))]))
        f.seek(btree)
        f.write(b"TREE" + bytes([0, 0]) + (1).to_bytes(2, "little") + u8(UNDEF) + u8(UNDEF) + u8(0) + u8(snod) + u8(name_off[names[-1]]))
        f.seek(snod)
        f.write(b"SNOD" + bytes([1, 0]) + len(names).to_bytes(2, "little"))
        for k in names:
            f.write(u8(name_off[k]) + u8(hdr_at[k]) + bytes(4) + bytes(4) + bytes(16))
        f.seek(heap)
        f.write(b"HEAP" + bytes(4) + u8(len(heap_data)) + u8(1) + u8(heap_seg))     # free-list head 1 = H5HL_FREE_NULL: no free block
        f.seek(heap_seg)
        f.write(bytes(heap_data))
        for k in names:
            space, dtm = hdrs[k]
            layout = bytes([3, 1]) + u8(data_at[k]) + u8(arrs[k].nbytes)
            f.seek(hdr_at[k])
            f.write(_v1_header([_v1_message(0x0001, space), _v1_message(0x0003, dtm), _v1_message(0x0008, layout)]))
        for k in names:
            f.seek(data_at[k])
            f.write(arrs[k].tobytes())
        f.truncate(eof)
    return path
