"""A minimal read-only HDF5 parser: enough to open the index files the reference writes.

The reference stores ``OnDiskIndex`` in HDF5 through h5py (``src/fast_forward/index/disk.py``):
root attributes ``num_vectors`` / ``ff_version`` (``:83-85``), chunked datasets ``vectors``,
``doc_ids``, ``psg_ids`` (``:145-165``) and an optional ``quantizer/{meta,attributes,data}`` group
tree (``:123-136``).  Neither h5py nor libhdf5 exists for this interpreter, and the hot path only
ever needs *where the raw chunks are* -- the reference itself memory-maps them by byte offset
(``disk.py:100-121``).  This module therefore parses just the classic ("earliest" libver, what
h5py writes by default) on-disk structures:

    superblock v0/v1 - symbol-table groups (B-tree v1 + local heap + SNOD) - object headers v1 with
    continuation blocks - dataspace / datatype / layout v3 (contiguous, chunked via B-tree v1) -
    attributes v1-v3 incl. variable-length strings through the global heap - enum (bool).

Anything else (filters, compression, new-style groups, layout v4) raises ``NotImplementedError``
with the feature's name.  Checked against files produced by real HDF5 (tests/golden/h5, written by
h5py 3.3.0 / HDF5 1.10.6 replaying the reference's calls: oracle/gen_h5_fixture.py).
"""

from __future__ import annotations

import struct
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


def is_hdf5(path: Path) -> bool:
    with open(path, "rb") as f:
        return f.read(8) == SIGNATURE


@dataclass
class Dataset:
    """Where a dataset's bytes are: one extent (contiguous) or many (chunked)."""

    shape: tuple[int, ...]
    dtype: np.dtype
    chunk_shape: tuple[int, ...] | None = None
    # chunked: {chunk origin (tuple of element offsets): (file address, stored bytes)}
    chunks: dict[tuple[int, ...], tuple[int, int]] = field(default_factory=dict)
    address: int | None = None  # contiguous: file address (None if never written)
    inline: bytes | None = None  # compact layout

    def read(self, path: Path) -> np.ndarray:
        """The whole dataset as an array (fill value 0 for chunks never written)."""
        if self.inline is not None:
            return np.frombuffer(self.inline, dtype=self.dtype).reshape(self.shape).copy()
        out = np.zeros(self.shape, dtype=self.dtype)
        if out.size == 0:
            return out
        with open(path, "rb") as f:
            if self.chunk_shape is None:
                if self.address is not None:
                    f.seek(self.address)
                    out = np.frombuffer(f.read(out.nbytes), dtype=self.dtype).reshape(self.shape).copy()
                return out
            n_chunk = int(np.prod(self.chunk_shape))
            for origin, (addr, size) in self.chunks.items():
                if size != n_chunk * self.dtype.itemsize:
                    raise NotImplementedError("HDF5: filtered (compressed) chunks")
                f.seek(addr)
                block = np.frombuffer(f.read(size), dtype=self.dtype).reshape(self.chunk_shape)
                sel = tuple(slice(o, min(o + c, s)) for o, c, s in zip(origin, self.chunk_shape, self.shape))
                if any(s.start >= s.stop for s in sel):
                    continue  # chunk beyond the current extent (after a shrink)
                out[sel] = block[tuple(slice(0, s.stop - s.start) for s in sel)]
        return out

    def read_rows(self, path: Path, lo: int, hi: int) -> np.ndarray:
        """Elements ``lo .. hi`` of a ONE-dimensional dataset (the id columns of an index file) without reading the rest:
        the extent of a contiguous dataset, the chunks that overlap the range of a chunked one (fill value 0 elsewhere)."""
        if len(self.shape) != 1:
            raise ValueError("read_rows is for one-dimensional datasets")
        lo, hi = max(0, int(lo)), min(int(hi), self.shape[0])
        if hi <= lo:
            return np.zeros(0, dtype=self.dtype)
        if self.inline is not None:
            return np.frombuffer(self.inline, dtype=self.dtype)[lo:hi].copy()
        item = self.dtype.itemsize
        out = np.zeros(hi - lo, dtype=self.dtype)
        with open(path, "rb") as f:
            if self.chunk_shape is None:
                if self.address is not None:
                    f.seek(self.address + lo * item)
                    out = np.frombuffer(f.read((hi - lo) * item), dtype=self.dtype).copy()
                return out
            per = int(self.chunk_shape[0])
            for c in range(lo // per, (hi - 1) // per + 1):
                where = self.chunks.get((c * per,))
                if where is None:
                    continue
                addr, size = where
                if size != per * item:
                    raise NotImplementedError("HDF5: filtered (compressed) chunks")
                a, b = max(lo, c * per), min(hi, (c + 1) * per)
                f.seek(addr + (a - c * per) * item)
                out[a - lo:b - lo] = np.frombuffer(f.read((b - a) * item), dtype=self.dtype)
        return out


@dataclass
class Group:
    attrs: dict
    datasets: dict[str, Dataset]
    groups: dict[str, "Group"]


class _Reader:
    def __init__(self, path: Path):
        self.path = Path(path)
        self.b = self.path.read_bytes() if self.path.stat().st_size < (64 << 20) else None
        self._f = None if self.b is not None else open(self.path, "rb")
        head = self._at(0, 64)
        if head[:8] != SIGNATURE:
            raise ValueError(f"{path} is not an HDF5 file")
        ver = head[8]
        if ver > 1:
            raise NotImplementedError(f"HDF5: superblock version {ver} (file written with a non-default libver)")
        self.O, self.L = head[13], head[14]
        if (self.O, self.L) != (8, 8):
            raise NotImplementedError("HDF5: offsets/lengths other than 8 bytes")
        pos = 24 + (4 if ver == 1 else 0)
        self.base = self._u64(pos)
        root_entry = pos + 4 * 8
        self.root_header = self._u64(root_entry + 8)

    # ---- raw access ------------------------------------------------------------------------------
    def _at(self, off: int, n: int) -> bytes:
        if self.b is not None:
            return self.b[off:off + n]
        self._f.seek(off)
        return self._f.read(n)

    def _u(self, off: int, fmt: str):
        return struct.unpack_from("<" + fmt, self._at(off, struct.calcsize("<" + fmt)))

    def _u64(self, off: int) -> int:
        return self._u(off, "Q")[0]

    # ---- object headers -----------------------------------------------------------------------------
    def messages(self, addr: int) -> list[tuple[int, bytes]]:
        """(type, body) of every message of a version-1 object header, following continuations."""
        if self._at(addr, 4) == b"OHDR":
            raise NotImplementedError("HDF5: version-2 object headers (file written with libver='latest')")
        version, _, n_msgs, _, size = self._u(addr, "BBHII")
        if version != 1:
            raise NotImplementedError(f"HDF5: object header version {version}")
        out: list[tuple[int, bytes]] = []
        blocks = [(addr + 16, size)]
        while blocks and len(out) < n_msgs:
            pos, left = blocks.pop(0)
            end = pos + left
            while pos + 8 <= end and len(out) < n_msgs:
                mtype, msize, _flags = self._u(pos, "HHB")
                body = self._at(pos + 8, msize)
                pos += 8 + msize
                if mtype == 0x10:  # continuation
                    c_off, c_len = struct.unpack_from("<QQ", body)
                    blocks.append((c_off + self.base, c_len))
                out.append((mtype, body))
        return out

    # ---- groups (symbol tables) -----------------------------------------------------------------------
    def _heap_string(self, heap_addr: int, offset: int) -> str:
        if self._at(heap_addr, 4) != b"HEAP":
            raise ValueError("HDF5: bad local heap")
        data_addr = self._u64(heap_addr + 8 + 2 * self.L)
        start = data_addr + self.base + offset
        raw = self._at(start, 512)
        return raw[:raw.index(b"\0")].decode()

    def _group_entries(self, btree: int, heap: int) -> dict[str, int]:
        """name -> object header address, walking the group B-tree down to its SNOD leaves."""
        out: dict[str, int] = {}
        sig = self._at(btree, 4)
        if sig == b"SNOD":
            _, _, n = self._u(btree + 4, "BBH")
            pos = btree + 8
            for _ in range(n):
                name_off, hdr = self._u(pos, "QQ")
                out[self._heap_string(heap, name_off)] = hdr + self.base
                pos += 2 * self.O + 24
            return out
        if sig != b"TREE":
            raise ValueError("HDF5: bad group B-tree node")
        ntype, _level, used = self._u(btree + 4, "BBH")
        if ntype != 0:
            raise ValueError("HDF5: expected a group B-tree")
        pos = btree + 8 + 2 * self.O
        for _ in range(used):
            pos += self.L  # key
            child = self._u64(pos)
            pos += self.O
            out.update(self._group_entries(child + self.base, heap))
        return out

    # ---- message decoders --------------------------------------------------------------------------------
    def _dataspace(self, body: bytes) -> tuple[int, ...]:
        ver, rank, flags = body[0], body[1], body[2]
        pos = 8 if ver == 1 else 4
        if ver not in (1, 2):
            raise NotImplementedError(f"HDF5: dataspace version {ver}")
        return tuple(struct.unpack_from(f"<{rank}Q", body, pos)) if rank else ()

    def _datatype(self, body: bytes) -> tuple[np.dtype | str, int]:
        """(numpy dtype or a tag for special classes, size in bytes)"""
        cls, bits0 = body[0] & 0x0F, body[1]
        size = struct.unpack_from("<I", body, 4)[0]
        if cls == 0:  # fixed point
            signed = bool(bits0 & 0x08)
            return np.dtype(f"{'<' if not bits0 & 1 else '>'}{'i' if signed else 'u'}{size}"), size
        if cls == 1:  # floating point
            return np.dtype(f"{'<' if not bits0 & 1 else '>'}f{size}"), size
        if cls == 3:  # fixed-length string
            return np.dtype(f"S{size}"), size
        if cls == 8:  # enum (h5py stores numpy bool_ as an int8 enum {FALSE, TRUE})
            base, _ = self._datatype(body[8:])
            return base, size
        if cls == 9:  # variable length
            if (bits0 & 0x0F) == 1:
                return "vlen_str", size
            raise NotImplementedError("HDF5: variable-length sequences")
        raise NotImplementedError(f"HDF5: datatype class {cls}")

    def _global_heap_object(self, collection: int, index: int) -> bytes:
        addr = collection + self.base
        if self._at(addr, 4) != b"GCOL":
            raise ValueError("HDF5: bad global heap collection")
        size = self._u64(addr + 8)
        pos, end = addr + 16, addr + size
        while pos + 16 <= end:
            idx, _, _, osize = self._u(pos, "HHIQ")
            if idx == 0:
                break
            if idx == index:
                return self._at(pos + 16, osize)
            pos += 16 + ((osize + 7) & ~7)
        raise ValueError("HDF5: global heap object not found")

    def _attribute(self, body: bytes):
        ver = body[0]
        name_sz, dt_sz, ds_sz = struct.unpack_from("<HHH", body, 2)
        pos = 8 + (1 if ver == 3 else 0)
        pad = (lambda n: (n + 7) & ~7) if ver == 1 else (lambda n: n)
        name = body[pos:pos + name_sz].split(b"\0")[0].decode()
        pos += pad(name_sz)
        dtype, size = self._datatype(body[pos:pos + dt_sz])
        pos += pad(dt_sz)
        shape = self._dataspace(body[pos:pos + ds_sz])
        pos += pad(ds_sz)
        count = int(np.prod(shape)) if shape else 1
        if isinstance(dtype, str):  # variable-length strings: (length, collection address, index) each
            vals = []
            for i in range(count):
                _length, coll, idx = struct.unpack_from("<IQI", body, pos + 16 * i)
                vals.append(self._global_heap_object(coll, idx).decode())
            value = vals[0] if not shape else np.array(vals, dtype=object).reshape(shape)
        else:
            arr = np.frombuffer(body, dtype=dtype, count=count, offset=pos)
            if dtype.kind == "S":
                value = arr[0].decode() if not shape else arr.reshape(shape)
            else:
                value = arr[0].item() if not shape else arr.reshape(shape).copy()
        return name, value

    def _chunk_btree(self, addr: int, rank: int, out: dict) -> None:
        if self._at(addr, 4) != b"TREE":
            raise ValueError("HDF5: bad chunk B-tree node")
        ntype, level, used = self._u(addr + 4, "BBH")
        if ntype != 1:
            raise ValueError("HDF5: expected a raw-data chunk B-tree")
        key_bytes = 8 + 8 * (rank + 1)
        pos = addr + 8 + 2 * self.O
        for _ in range(used):
            size, mask = self._u(pos, "II")
            origin = struct.unpack_from(f"<{rank}Q", self._at(pos + 8, 8 * rank))
            child = self._u64(pos + key_bytes)
            pos += key_bytes + self.O
            if level > 0:
                self._chunk_btree(child + self.base, rank, out)
            else:
                if mask:
                    raise NotImplementedError("HDF5: filtered chunks")
                out[tuple(origin)] = (child + self.base, size)

    def _dataset(self, msgs: list[tuple[int, bytes]]) -> Dataset:
        shape, dtype, ds = None, None, None
        for mtype, body in msgs:
            if mtype == 0x01:
                shape = self._dataspace(body)
            elif mtype == 0x03:
                dtype, _ = self._datatype(body)
                if isinstance(dtype, str):
                    raise NotImplementedError("HDF5: variable-length dataset")
            elif mtype == 0x0B:
                raise NotImplementedError("HDF5: filter pipeline (compressed dataset)")
        for mtype, body in msgs:
            if mtype != 0x08:
                continue
            ver, cls = body[0], body[1]
            if ver != 3:
                raise NotImplementedError(f"HDF5: data layout version {ver}")
            if cls == 0:
                n = struct.unpack_from("<H", body, 2)[0]
                ds = Dataset(shape, dtype, inline=bytes(body[4:4 + n]))
            elif cls == 1:
                addr = struct.unpack_from("<Q", body, 2)[0]
                ds = Dataset(shape, dtype, address=None if addr == _UNDEF else addr + self.base)
            elif cls == 2:
                rank1 = body[2]
                btree = struct.unpack_from("<Q", body, 3)[0]
                dims = struct.unpack_from(f"<{rank1}I", body, 11)
                ds = Dataset(shape, dtype, chunk_shape=tuple(dims[:-1]))
                if btree != _UNDEF:
                    self._chunk_btree(btree + self.base, rank1 - 1, ds.chunks)
            else:
                raise NotImplementedError(f"HDF5: layout class {cls}")
        if ds is None or shape is None or dtype is None:
            raise ValueError("HDF5: incomplete dataset header")
        return ds

    def group(self, header_addr: int) -> Group:
        msgs = self.messages(header_addr)
        attrs = dict(self._attribute(body) for mtype, body in msgs if mtype == 0x0C)
        g = Group(attrs=attrs, datasets={}, groups={})
        for mtype, body in msgs:
            if mtype in (0x02, 0x06):
                raise NotImplementedError("HDF5: new-style (link message) groups")
            if mtype == 0x11:  # symbol table: B-tree + local heap
                btree, heap = struct.unpack_from("<QQ", body)
                for name, hdr in self._group_entries(btree + self.base, heap + self.base).items():
                    child = self.messages(hdr)
                    if any(t == 0x08 for t, _ in child):
                        g.datasets[name] = self._dataset(child)
                        # dataset attributes are not needed by the index format
                    else:
                        g.groups[name] = self.group(hdr)
        return g


def read(path: Path) -> Group:
    """Parse an HDF5 file into nested groups with attributes and dataset locations."""
    r = _Reader(Path(path))
    try:
        return r.group(r.root_header + r.base)
    finally:
        if r._f is not None:
            r._f.close()
