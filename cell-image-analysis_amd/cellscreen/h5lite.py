"""A minimal pure-Python reader for the subset of HDF5 that `model.weights.h5` inside a `.keras`
archive uses: groups (old-style symbol tables, or new-style compact link messages) and contiguous /
compact datasets of little-endian IEEE floats and integers.  No chunking, compression, dense link
storage, references or attributes -- h5py writes none of those for plain `f[name] = array` stores,
which is all Keras's H5IOStore does.  Exists because neither h5py nor Keras is installed where this
package runs (the reference loads `best_autoencoder.keras` / `encoder.keras` with Keras,
improved_detection.py:28-29); layout per the HDF5 File Format Specification v3.

    tree = h5lite.read(bytes_or_path)      # {"layers/conv2d/vars/0": ndarray, ...}
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(ValueError):
    pass


class _File:
    def __init__(self, buf: bytes):
        self.b = buf
        base = -1
        off = 0
        while off < len(buf):                       # the superblock sits at 0, 512, 1024, ...
            if buf[off:off + 8] == SIGNATURE:
                base = off
                break
            off = 512 if off == 0 else off * 2
        if base < 0:
            raise H5Error("not an HDF5 file (signature not found)")
        self.base = base
        ver = buf[base + 8]
        if ver in (0, 1):
            self.so, self.sl = buf[base + 13], buf[base + 14]
            p = base + 24 + (4 if ver == 1 else 0)
            p += 4 * self.so                        # base address, free-space, end of file, driver info
            # root group symbol table entry: link name offset, object header address, cache type, reserved, scratch
            self.root = self.u(p + self.so, self.so)
        elif ver in (2, 3):
            self.so, self.sl = buf[base + 9], buf[base + 10]
            p = base + 12 + 3 * self.so             # base address, superblock extension, end of file
            self.root = self.u(p, self.so)
        else:
            raise H5Error(f"superblock version {ver} not supported")
        if self.so != 8 or self.sl != 8:
            raise H5Error("only 8-byte offsets/lengths are supported")

    def u(self, off: int, n: int) -> int:
        return int.from_bytes(self.b[off:off + n], "little")

    # ---- object headers ---------------------------------------------------------------------
    def messages(self, addr: int):
        """Yields (type, bytes) for every header message of the object at `addr` (v1 and v2 headers)."""
        b = self.b
        a = self.base + addr
        if b[a:a + 4] == b"OHDR":
            yield from self._messages_v2(a)
            return
        if b[a] != 1:
            raise H5Error(f"object header version {b[a]} at {addr:#x} not supported")
        nmsg = self.u(a + 2, 2)
        size = self.u(a + 8, 4)
        blocks = [(a + 16, size)]                   # the first block starts after 4 bytes of alignment padding
        seen = 0
        while blocks and seen < nmsg:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and seen < nmsg:
                mtype, msize = self.u(p, 2), self.u(p + 2, 2)
                data = b[p + 8:p + 8 + msize]
                p += 8 + msize
                seen += 1
                if mtype == 0x0010:                 # continuation
                    blocks.append((self.base + self.u_b(data, 0, 8), self.u_b(data, 8, 8)))
                else:
                    yield mtype, data

    def _messages_v2(self, a: int):
        b = self.b
        flags = b[a + 5]
        p = a + 6
        if flags & 0x20:
            p += 16
        if flags & 0x10:
            p += 4
        nsz = 1 << (flags & 3)
        chunk = self.u(p, nsz)
        p += nsz
        blocks = [(p, chunk)]
        while blocks:
            p, left = blocks.pop(0)
            end = p + left
            while p + 4 <= end:
                mtype, msize, _mflags = b[p], self.u(p + 1, 2), b[p + 3]
                p += 4 + (2 if flags & 4 else 0)
                data = b[p:p + msize]
                p += msize
                if mtype == 0x10:
                    ca, cl = self.base + self.u_b(data, 0, 8), self.u_b(data, 8, 8)
                    if b[ca:ca + 4] != b"OCHK":
                        raise H5Error("bad object header continuation")
                    blocks.append((ca + 4, cl - 8))   # minus signature and checksum
                elif mtype != 0:
                    yield mtype, data

    @staticmethod
    def u_b(data: bytes, off: int, n: int) -> int:
        return int.from_bytes(data[off:off + n], "little")

    # ---- groups ---------------------------------------------------------------------------------
    def children(self, addr: int) -> Optional[Dict[str, int]]:
        """name -> object header address, or None if the object is not a group."""
        out: Dict[str, int] = {}
        is_group = False
        for mtype, data in self.messages(addr):
            if mtype == 0x0011:                     # symbol table: B-tree + local heap
                is_group = True
                self._walk_btree(self.u_b(data, 0, 8), self._heap_data(self.u_b(data, 8, 8)), out)
            elif mtype == 0x0002:                   # link info
                is_group = True
                ver, lflags = data[0], data[1]
                p = 2 + (8 if lflags & 1 else 0)
                if self.u_b(data, p, 8) != UNDEF:
                    raise H5Error("dense link storage (fractal heap) is not supported")
            elif mtype == 0x0006:                   # link
                is_group = True
                name, target = self._link(data)
                if target is not None:
                    out[name] = target
        return out if is_group else None

    def _link(self, data: bytes):
        if data[0] != 1:
            raise H5Error("link message version")
        flags = data[1]
        p = 2
        ltype = 0
        if flags & 0x08:
            ltype = data[p]; p += 1
        if flags & 0x04:
            p += 8
        if flags & 0x10:
            p += 1
        nsz = 1 << (flags & 3)
        nlen = self.u_b(data, p, nsz); p += nsz
        name = data[p:p + nlen].decode("utf-8"); p += nlen
        if ltype != 0:
            return name, None                       # soft / external links: ignored
        return name, self.u_b(data, p, 8)

    def _heap_data(self, addr: int) -> int:
        a = self.base + addr
        if self.b[a:a + 4] != b"HEAP":
            raise H5Error("bad local heap")
        return self.base + self.u(a + 8 + 2 * self.sl, self.so)

    def _walk_btree(self, addr: int, heap: int, out: Dict[str, int]):
        a = self.base + addr
        b = self.b
        if b[a:a + 4] != b"TREE" or b[a + 4] != 0:
            raise H5Error("bad group B-tree node")
        level, used = b[a + 5], self.u(a + 6, 2)
        p = a + 8 + 2 * self.so                      # skip sibling addresses
        for i in range(used):
            child = self.u(p + self.sl + i * (self.sl + self.so), self.so)
            if level > 0:
                self._walk_btree(child, heap, out)
            else:
                s = self.base + child
                if b[s:s + 4] != b"SNOD":
                    raise H5Error("bad symbol table node")
                n = self.u(s + 6, 2)
                e = s + 8
                for _ in range(n):
                    name_off, obj = self.u(e, self.so), self.u(e + self.so, self.so)
                    z = b.index(b"\0", heap + name_off)
                    out[b[heap + name_off:z].decode("utf-8")] = obj
                    e += 2 * self.so + 24

    # ---- datasets -------------------------------------------------------------------------------
    def dataset(self, addr: int) -> Optional[np.ndarray]:
        shape = dtype = None
        layout = None
        for mtype, data in self.messages(addr):
            if mtype == 0x0001:
                ver, rank = data[0], data[1]
                p = 8 if ver == 1 else 4
                shape = tuple(self.u_b(data, p + 8 * i, 8) for i in range(rank))
            elif mtype == 0x0003:
                cls, bits0, size = data[0] & 0x0F, data[1], self.u_b(data, 4, 4)
                if bits0 & 1:
                    raise H5Error("big-endian data is not supported")
                if cls == 1 and size in (2, 4, 8):
                    dtype = np.dtype(f"<f{size}")
                elif cls == 0 and size in (1, 2, 4, 8):
                    dtype = np.dtype(("<i" if data[1] & 0x08 else "<u") + str(size))
                else:
                    raise H5Error(f"datatype class {cls} size {size} is not supported")
            elif mtype == 0x0008:
                ver = data[0]
                if ver not in (3, 4):                   # 4 differs from 3 only for chunked storage
                    raise H5Error(f"data layout version {ver} is not supported")
                lc = data[1]
                if lc == 0:
                    n = self.u_b(data, 2, 2)
                    layout = ("compact", data[4:4 + n])
                elif lc == 1:
                    layout = ("contiguous", self.u_b(data, 2, 8), self.u_b(data, 10, 8))
                else:
                    raise H5Error("chunked datasets are not supported (Keras writes contiguous ones)")
        if shape is None or dtype is None or layout is None:
            return None
        count = int(np.prod(shape)) if shape else 1
        if layout[0] == "compact":
            raw = layout[1]
        else:
            if layout[1] == UNDEF:
                return np.zeros(shape, dtype)
            raw = self.b[self.base + layout[1]:self.base + layout[1] + layout[2]]
        return np.frombuffer(raw, dtype=dtype, count=count).reshape(shape).copy()


def read(src) -> Dict[str, np.ndarray]:
    """All datasets of the file as {"group/sub/name": array}.  `src`: bytes or a path."""
    if not isinstance(src, (bytes, bytearray, memoryview)):
        with open(src, "rb") as f:
            src = f.read()
    f = _File(bytes(src))
    out: Dict[str, np.ndarray] = {}

    def walk(addr: int, prefix: str, depth: int):
        if depth > 32:
            raise H5Error("group nesting too deep")
        kids = f.children(addr)
        if kids is None:
            d = f.dataset(addr)
            if d is not None:
                out[prefix] = d
            return
        for name, target in kids.items():
            walk(target, f"{prefix}/{name}" if prefix else name, depth + 1)

    walk(f.root, "", 0)
    return out


# =================================================================================================
# Writer: the same subset, in the "earliest" layout h5py (and therefore Keras's H5IOStore) produces --
# version-0 superblock, version-1 object headers, symbol-table groups (local heap + one level-0 B-tree
# node + symbol-table nodes of 2 x leafK = 8 entries), contiguous little-endian datasets.  Written so
# that the trainer can emit `best_autoencoder.keras` / `final_autoencoder.keras` / `encoder.keras`
# (CAE_improved_modeltrain.py:271,299-300) without h5py or Keras; tests open the result with the real
# HDF5 library (conda h5py) as well as with the reader above.
# =================================================================================================
_LEAF_K, _INTERNAL_K = 4, 16
_H5HL_FREE_NULL = 1                     # "end of free list" sentinel of the local heap (H5HLprivate.h)


def _pad8(b: bytes) -> bytes:
    return b + b"\0" * ((8 - len(b) % 8) % 8)


def _msg(mtype: int, data: bytes, flags: int = 0) -> bytes:
    data = _pad8(data)
    return mtype.to_bytes(2, "little") + len(data).to_bytes(2, "little") + bytes([flags, 0, 0, 0]) + data


def _object_header(msgs) -> bytes:
    body = b"".join(msgs)
    # version 1, reserved, #messages, reference count 1, header size; 4 bytes of padding align the messages to 8
    return bytes([1, 0]) + len(msgs).to_bytes(2, "little") + (1).to_bytes(4, "little") + len(body).to_bytes(4, "little") + b"\0" * 4 + body


def _datatype_msg(dt: np.dtype) -> bytes:
    dt = np.dtype(dt)
    if dt.kind == "f" and dt.itemsize in (4, 8):
        size = dt.itemsize
        expb, manb, bias = (8, 23, 127) if size == 4 else (11, 52, 1023)
        head = bytes([0x11, 0x20, 8 * size - 1, 0]) + size.to_bytes(4, "little")       # class 1 (float) v1; LE, implied-msb mantissa; sign bit
        props = (0).to_bytes(2, "little") + (8 * size).to_bytes(2, "little") + bytes([manb, expb, 0, manb]) + bias.to_bytes(4, "little")
        return head + props
    if dt.kind in "iu" and dt.itemsize in (1, 2, 4, 8):
        size = dt.itemsize
        head = bytes([0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0]) + size.to_bytes(4, "little")   # class 0 (fixed point) v1; LE; signed bit
        return head + (0).to_bytes(2, "little") + (8 * size).to_bytes(2, "little")
    raise H5Error(f"dtype {dt} is not supported by the writer")


class _Writer:
    def __init__(self):
        self.buf = bytearray(b"\0" * 96)              # superblock, filled in last

    def alloc(self, data: bytes) -> int:
        self.buf += b"\0" * ((8 - len(self.buf) % 8) % 8)
        addr = len(self.buf)
        self.buf += data
        return addr

    def dataset(self, arr: np.ndarray) -> int:
        a = np.asarray(arr)
        if not a.flags.c_contiguous:
            a = a.copy(order="C")                     # (np.ascontiguousarray would turn a scalar into shape (1,))
        if a.dtype.byteorder == ">":
            a = a.astype(a.dtype.newbyteorder("<"))
        raw = a.tobytes()
        data_addr = self.alloc(raw) if raw else UNDEF
        space = bytes([1, a.ndim, 0, 0, 0, 0, 0, 0]) + b"".join(int(d).to_bytes(8, "little") for d in a.shape)
        fill = bytes([2, 2, 2, 0])                    # version 2; allocate late; write fill value if set; none defined
        layout = bytes([3, 1]) + data_addr.to_bytes(8, "little") + len(raw).to_bytes(8, "little")
        return self.alloc(_object_header([_msg(0x0001, space), _msg(0x0003, _datatype_msg(a.dtype), flags=1), _msg(0x0005, fill),
                                          _msg(0x0008, layout)]))

    def group(self, tree: dict):
        """tree: {name: ndarray | dict} -> (object header address, B-tree address, heap address)."""
        names = sorted(tree, key=lambda s: s.encode())                     # symbol-table order = strcmp order
        if len(names) > 2 * _LEAF_K * 2 * _INTERNAL_K:
            raise H5Error("too many links in one group for a single B-tree node")
        kids = {}
        for n in names:
            v = tree[n]
            kids[n] = self.group(v) if isinstance(v, dict) else (self.dataset(np.asarray(v)), None, None)
        # local heap: "" at offset 0, then the names, each NUL-terminated and padded to 8
        seg = bytearray(b"\0" * 8)
        off = {}
        for n in names:
            off[n] = len(seg)
            seg += _pad8(n.encode() + b"\0")
        seg_addr = self.alloc(bytes(seg))
        heap = b"HEAP" + bytes(4) + len(seg).to_bytes(8, "little") + _H5HL_FREE_NULL.to_bytes(8, "little") + seg_addr.to_bytes(8, "little")
        heap_addr = self.alloc(heap)
        # symbol-table nodes of up to 2 x leafK entries, one level-0 B-tree node over them
        keys, children = [0], []
        per = 2 * _LEAF_K
        for i in range(0, len(names), per):
            part = names[i:i + per]
            node = bytearray(b"SNOD" + bytes([1, 0]) + len(part).to_bytes(2, "little"))
            for n in part:
                hdr, bt, hp = kids[n]
                node += off[n].to_bytes(8, "little") + hdr.to_bytes(8, "little")
                if bt is not None:
                    node += (1).to_bytes(4, "little") + bytes(4) + bt.to_bytes(8, "little") + hp.to_bytes(8, "little")
                else:
                    node += bytes(4) + bytes(4) + bytes(16)
            node += bytes(8 + per * 40 - len(node))
            children.append(self.alloc(bytes(node)))
            keys.append(off[part[-1]])
        bt = bytearray(b"TREE" + bytes([0, 0]) + len(children).to_bytes(2, "little") + UNDEF.to_bytes(8, "little") * 2)
        for i, c in enumerate(children):
            bt += keys[i].to_bytes(8, "little") + c.to_bytes(8, "little")
        bt += keys[len(children)].to_bytes(8, "little")
        bt += bytes(24 + (2 * _INTERNAL_K + 1) * 8 + 2 * _INTERNAL_K * 8 - len(bt))
        bt_addr = self.alloc(bytes(bt))
        hdr_addr = self.alloc(_object_header([_msg(0x0011, bt_addr.to_bytes(8, "little") + heap_addr.to_bytes(8, "little"))]))
        return hdr_addr, bt_addr, heap_addr

    def finish(self, root) -> bytes:
        hdr, bt, hp = root
        self.buf += b"\0" * ((8 - len(self.buf) % 8) % 8)
        sb = bytearray(SIGNATURE)
        sb += bytes([0, 0, 0, 0, 0, 8, 8, 0])
        sb += _LEAF_K.to_bytes(2, "little") + _INTERNAL_K.to_bytes(2, "little") + bytes(4)
        sb += (0).to_bytes(8, "little") + UNDEF.to_bytes(8, "little") + len(self.buf).to_bytes(8, "little") + UNDEF.to_bytes(8, "little")
        sb += (0).to_bytes(8, "little") + hdr.to_bytes(8, "little") + (1).to_bytes(4, "little") + bytes(4) + bt.to_bytes(8, "little") + hp.to_bytes(8, "little")
        assert len(sb) == 96
        self.buf[0:96] = sb
        return bytes(self.buf)


def write(tree: dict) -> bytes:
    """{"group/sub/name": array} (flat, as read() returns) or nested dicts -> the bytes of an HDF5 file.  Empty groups are
    written for keys whose value is an empty dict."""
    nested: dict = {}
    for key, v in tree.items():
        parts = [p for p in key.split("/") if p]
        d = nested
        for p in parts[:-1]:
            d = d.setdefault(p, {})
            if not isinstance(d, dict):
                raise H5Error(f"{key}: a dataset is in the way")
        if isinstance(v, dict):
            sub = d.setdefault(parts[-1], {})
            for k2, v2 in (write_flatten(v)).items():
                sub[k2] = v2
        else:
            d[parts[-1]] = v
    w = _Writer()
    return w.finish(w.group(nested))


def write_flatten(d: dict) -> dict:
    """Nested dicts stay nested (write() accepts both forms); helper kept separate for clarity."""
    return d
