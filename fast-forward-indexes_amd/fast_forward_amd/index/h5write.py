"""A minimal HDF5 writer: enough to produce the index files the reference reads.

Counterpart of :mod:`h5read`.  ``OnDiskIndex.load`` of the reference (``index/disk.py:355-418``)
opens a file through h5py and expects: root attribute ``num_vectors``; chunked, unfiltered,
resizable datasets ``vectors`` ``(capacity, dim)``, ``doc_ids`` / ``psg_ids`` ``(capacity,)`` of
fixed-width byte strings (``""`` = no id); optionally ``quantizer/{meta,attributes}`` (attributes
only) and ``quantizer/data`` (one dataset per array).  No h5py or libhdf5 exists for this
interpreter, so the file is laid out by hand in the classic ("earliest" libver) structures that
every HDF5 release reads:

    superblock v0 - symbol-table groups (B-tree v1 + local heap + SNOD) - object headers v1 -
    dataspace v1 / datatype v1 / fill value v2 / layout v3 (contiguous, or chunked with a B-tree v1
    of default fan-out, as many levels as the chunk count needs) - attribute v1, strings as
    variable-length UTF-8 through one global heap collection (h5py then returns ``str``),
    booleans as h5py's enum.

Raw chunk data is appended first (streamed, so an index larger than memory can be written), all
metadata goes behind it, and the superblock at offset 0 is filled in last.  Files are checked in
this container with real HDF5 (h5py 3.3.0 / HDF5 1.10.6 under /opt/conda, tests/test_ondisk.py)
and with :mod:`h5read`.
"""

from __future__ import annotations

import struct
from collections.abc import Iterable
from pathlib import Path

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF
_GROUP_LEAF_K, _GROUP_INTERNAL_K, _CHUNK_K = 4, 16, 32  # library defaults (superblock v0 has no chunk K field)
_SUPERBLOCK_BYTES = 96


def _pad8(b: bytes) -> bytes:
    return b + b"\0" * (-len(b) % 8)


# ---- datatype / dataspace messages ---------------------------------------------------------------
def _float_type(size: int, exp_bits: int, mant_bits: int) -> bytes:
    sign = size * 8 - 1
    head = struct.pack("<BBBBI", 0x11, 0x20, sign, 0, size)  # v1 class 1; little-endian, implied mantissa msb
    return head + struct.pack("<HHBBBBI", 0, size * 8, mant_bits, exp_bits, 0, mant_bits, (1 << (exp_bits - 1)) - 1)


def _int_type(size: int, signed: bool) -> bytes:
    return struct.pack("<BBBBI", 0x10, 0x08 if signed else 0x00, 0, 0, size) + struct.pack("<HH", 0, size * 8)


def _fixed_string_type(size: int) -> bytes:
    return struct.pack("<BBBBI", 0x13, 0x01, 0, 0, size)  # null-padded ASCII, what h5py makes of numpy 'S<n>'


def _vlen_string_type() -> bytes:
    # variable-length "string" (type 1), null-terminated, UTF-8; the base type is a one-byte
    # integer -- byte for byte what h5py writes for a str attribute
    return struct.pack("<BBBBI", 0x19, 0x01, 0x01, 0, 16) + _int_type(1, False)


def _bool_type() -> bytes:
    names = _pad8(b"FALSE\0") + _pad8(b"TRUE\0")
    return struct.pack("<BBBBI", 0x18, 2, 0, 0, 1) + _int_type(1, True) + names + bytes([0, 1])  # enum {FALSE=0, TRUE=1} of int8


def datatype_message(dtype: np.dtype) -> bytes:
    dtype = np.dtype(dtype)
    if dtype.kind == "f" and dtype.itemsize in (2, 4, 8):
        return _float_type(dtype.itemsize, {2: 5, 4: 8, 8: 11}[dtype.itemsize], {2: 10, 4: 23, 8: 52}[dtype.itemsize])
    if dtype.kind in "iu":
        return _int_type(dtype.itemsize, dtype.kind == "i")
    if dtype.kind == "S":
        return _fixed_string_type(dtype.itemsize)
    raise NotImplementedError(f"HDF5 datatype for numpy dtype {dtype}")


def dataspace_message(shape: tuple[int, ...], maxshape: tuple[int | None, ...] | None = None) -> bytes:
    rank = len(shape)
    body = struct.pack("<BBB5x", 1, rank, 1 if maxshape is not None and rank else 0)
    body += b"".join(struct.pack("<Q", int(s)) for s in shape)
    if maxshape is not None and rank:
        body += b"".join(struct.pack("<Q", _UNDEF if m is None else int(m)) for m in maxshape)
    return body


# ---- the tree the caller builds ----------------------------------------------------------------
class _Node:
    def __init__(self) -> None:
        self.attrs: dict[str, object] = {}


class _Group(_Node):
    def __init__(self) -> None:
        super().__init__()
        self.children: dict[str, _Node] = {}


class _Dataset(_Node):
    def __init__(self, dtype: np.dtype, shape, maxshape, chunk_shape) -> None:
        super().__init__()
        self.dtype, self.shape, self.maxshape, self.chunk_shape = np.dtype(dtype), tuple(shape), maxshape, chunk_shape
        self.chunks: list[tuple[tuple[int, ...], int, int]] = []  # (offset per dim, file address, bytes)
        self.contiguous: tuple[int, int] | None = None          # (address, bytes)


class H5Writer:
    """Build a file: ``create_group`` / ``write_chunked`` / ``write_contiguous`` / ``set_attr`` in any
    order (raw data is written through immediately), then ``close``."""

    def __init__(self, path: Path):
        self._f = open(path, "wb")
        self._f.write(b"\0" * 4096)  # superblock goes here at the end; raw data starts page aligned
        self._pos = 4096
        self.root = _Group()
        self._heap_objects: list[bytes] = []  # global heap: variable-length strings of attributes

    # -- raw file space ---------------------------------------------------------------------------
    def _append(self, data, align: int = 8) -> int:
        pad = -self._pos % align
        if pad:
            self._f.write(b"\0" * pad)
            self._pos += pad
        addr = self._pos
        self._f.write(data)
        self._pos += len(data) if isinstance(data, (bytes, bytearray)) else data.nbytes
        return addr

    # -- building ---------------------------------------------------------------------------------
    def _parent(self, path: str, create: bool = True) -> tuple[_Group, str]:
        parts = [p for p in path.split("/") if p]
        g = self.root
        for p in parts[:-1]:
            nxt = g.children.get(p)
            if nxt is None:
                if not create:
                    raise KeyError(path)
                nxt = g.children[p] = _Group()
            if not isinstance(nxt, _Group):
                raise ValueError(f"{p} is a dataset")
            g = nxt
        return g, parts[-1]

    def create_group(self, path: str) -> _Group:
        parent, name = self._parent(path)
        node = parent.children.setdefault(name, _Group())
        assert isinstance(node, _Group)
        return node

    def node(self, path: str) -> _Node:
        if path.strip("/") == "":
            return self.root
        parent, name = self._parent(path, create=False)
        return parent.children[name]

    def set_attr(self, path: str, name: str, value) -> None:
        self.node(path).attrs[name] = value

    def write_chunked(self, path: str, dtype, capacity: int, trailing: tuple[int, ...], chunk_rows: int,
                      blocks: Iterable[np.ndarray]) -> None:
        """A resizable dataset ``(capacity, *trailing)`` in chunks of ``chunk_rows`` rows (unlimited
        along axis 0, like ``create_dataset(..., maxshape=(None, ...))``).  ``blocks`` yields the
        rows in order, any number of rows at a time; chunks past the last row are not allocated."""
        dtype = np.dtype(dtype)
        ds = _Dataset(dtype, (capacity, *trailing), (None, *trailing), (chunk_rows, *trailing))
        parent, name = self._parent(path)
        parent.children[name] = ds
        row_bytes = int(np.prod(trailing, dtype=np.int64)) * dtype.itemsize if trailing else dtype.itemsize
        pending = np.zeros((chunk_rows, *trailing), dtype=dtype)
        filled, row0 = 0, 0

        def flush() -> None:
            nonlocal filled, row0
            if filled == 0:
                return
            if filled < chunk_rows:
                pending[filled:] = np.zeros((), dtype=dtype)  # a partial last chunk is stored whole, zero filled
            addr = self._append(np.ascontiguousarray(pending), align=4096)
            ds.chunks.append(((row0, *([0] * len(trailing))), addr, chunk_rows * row_bytes))
            row0 += chunk_rows
            filled = 0

        for block in blocks:
            block = np.asarray(block, dtype=dtype).reshape(-1, *trailing)
            done = 0
            while done < block.shape[0]:
                take = min(chunk_rows - filled, block.shape[0] - done)
                pending[filled:filled + take] = block[done:done + take]
                filled += take
                done += take
                if filled == chunk_rows:
                    flush()
        flush()
        if row0 > capacity:
            raise ValueError("more rows written than the dataset's capacity")

    def write_contiguous(self, path: str, array: np.ndarray) -> None:
        array = np.ascontiguousarray(array)
        ds = _Dataset(array.dtype, array.shape, None, None)
        parent, name = self._parent(path)
        parent.children[name] = ds
        ds.contiguous = (self._append(array) if array.nbytes else _UNDEF, array.nbytes)

    # -- metadata ---------------------------------------------------------------------------------
    def _attribute_message(self, name: str, value) -> bytes:
        if isinstance(value, (bool, np.bool_)):
            dt, data = _bool_type(), bytes([1 if value else 0])
        elif isinstance(value, (int, np.integer)):
            dt, data = _int_type(8, True), struct.pack("<q", int(value))
        elif isinstance(value, (float, np.floating)):
            dt, data = _float_type(8, 11, 52), struct.pack("<d", float(value))
        elif isinstance(value, str):
            raw = value.encode("utf-8")
            self._heap_objects.append(raw)
            # (length, collection address, object index): the address is patched in close()
            dt, data = _vlen_string_type(), struct.pack("<IQI", len(raw), 0, len(self._heap_objects))
        else:
            raise NotImplementedError(f"HDF5 attribute of type {type(value).__name__}")
        name_b = name.encode("utf-8") + b"\0"
        space = dataspace_message(())
        return struct.pack("<BBHHH", 1, 0, len(name_b), len(dt), len(space)) + _pad8(name_b) + _pad8(dt) + _pad8(space) + data

    @staticmethod
    def _object_header(messages: list[tuple[int, bytes]]) -> bytes:
        body = b"".join(struct.pack("<HHB3x", mtype, len(_pad8(data)), 0) + _pad8(data) for mtype, data in messages)
        return struct.pack("<BBHII4x", 1, 0, len(messages), 1, len(body)) + body

    def _chunk_btree(self, ds: _Dataset) -> int:
        """B-tree v1 (node type 1) over the chunks, leaves first; returns the root's address."""
        rank = len(ds.shape)
        key_fmt = "<II" + "Q" * (rank + 1)
        key_bytes = struct.calcsize(key_fmt)
        node_bytes = 24 + (2 * _CHUNK_K + 1) * key_bytes + 2 * _CHUNK_K * 8

        def key(offset: tuple[int, ...], nbytes: int) -> bytes:
            return struct.pack(key_fmt, nbytes, 0, *offset, 0)

        past = lambda off: (off[0] + ds.chunk_shape[0], *off[1:])  # noqa: E731  (bounds the last child from above)
        # level 0 entries: (first key offset, chunk bytes, child address, last offset covered)
        entries = [(off, nbytes, addr, off) for off, addr, nbytes in ds.chunks]
        if not entries:
            return _UNDEF
        level = 0
        while True:
            groups = [entries[i:i + 2 * _CHUNK_K] for i in range(0, len(entries), 2 * _CHUNK_K)]
            base = self._pos + (-self._pos % 8)
            addrs = [base + i * (node_bytes + (-node_bytes % 8)) for i in range(len(groups))]
            nxt = []
            for gi, grp in enumerate(groups):
                left = addrs[gi - 1] if gi > 0 else _UNDEF
                right = addrs[gi + 1] if gi + 1 < len(groups) else _UNDEF
                node = b"TREE" + struct.pack("<BBHQQ", 1, level, len(grp), left, right)
                for off, nbytes, child, _ in grp:
                    node += key(off, nbytes) + struct.pack("<Q", child)
                node += key(past(grp[-1][3]), 0)
                node = node.ljust(node_bytes, b"\0")
                got = self._append(node)
                assert got == addrs[gi]
                nxt.append((grp[0][0], grp[0][1], got, grp[-1][3]))
            if len(nxt) == 1:
                return nxt[0][2]
            entries, level = nxt, level + 1

    def _write_dataset(self, ds: _Dataset) -> int:
        msgs = [(0x0001, dataspace_message(ds.shape, ds.maxshape)), (0x0003, datatype_message(ds.dtype))]
        if ds.chunk_shape is not None:
            msgs.append((0x0005, struct.pack("<BBBBI", 2, 3, 0, 1, 0)))  # fill value v2: incremental allocation, default (zero) fill
            layout = struct.pack("<BBB", 3, 2, len(ds.shape) + 1) + struct.pack("<Q", self._chunk_btree(ds))
            layout += b"".join(struct.pack("<I", int(c)) for c in ds.chunk_shape) + struct.pack("<I", ds.dtype.itemsize)
        else:
            msgs.append((0x0005, struct.pack("<BBBBI", 2, 2, 2, 1, 0)))  # late allocation, default fill
            layout = struct.pack("<BBQQ", 3, 1, ds.contiguous[0], ds.contiguous[1])
        msgs.append((0x0008, layout))
        msgs += [(0x000C, self._attribute_message(k, v)) for k, v in ds.attrs.items()]
        return self._append(self._object_header(msgs))

    def _write_group(self, g: _Group) -> tuple[int, int, int]:
        """Children first; returns (object header, B-tree, local heap) addresses."""
        if len(g.children) > 2 * _GROUP_LEAF_K:
            raise NotImplementedError("more than 8 links in one group")
        child_addr: dict[str, tuple[int, tuple[int, int] | None]] = {}
        for name, node in g.children.items():
            if isinstance(node, _Group):
                hdr, bt, heap = self._write_group(node)
                child_addr[name] = (hdr, (bt, heap))
            else:
                child_addr[name] = (self._write_dataset(node), None)
        # local heap: offset 0 is the empty string, then the link names
        names = sorted(child_addr, key=lambda s: s.encode("utf-8"))
        segment, name_off = bytearray(8), {}
        for n in names:
            name_off[n] = len(segment)
            segment += _pad8(n.encode("utf-8") + b"\0")
        seg_addr = self._append(bytes(segment))
        heap_addr = self._append(b"HEAP" + struct.pack("<B3xQQQ", 0, len(segment), 1, seg_addr))  # free list: none (H5HL_FREE_NULL)
        snod = b"SNOD" + struct.pack("<BBH", 1, 0, len(names))
        for n in names:
            hdr, sub = child_addr[n]
            if sub is None:
                snod += struct.pack("<QQII16x", name_off[n], hdr, 0, 0)
            else:
                snod += struct.pack("<QQIIQQ", name_off[n], hdr, 1, 0, sub[0], sub[1])
        snod_addr = self._append(snod.ljust(8 + 2 * _GROUP_LEAF_K * 40, b"\0"))
        node = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if names else 0, _UNDEF, _UNDEF)
        if names:
            node += struct.pack("<QQQ", 0, snod_addr, name_off[names[-1]])
        bt_addr = self._append(node.ljust(24 + (2 * _GROUP_INTERNAL_K + 1) * 8 + 2 * _GROUP_INTERNAL_K * 8, b"\0"))
        msgs = [(0x0011, struct.pack("<QQ", bt_addr, heap_addr))]
        msgs += [(0x000C, self._attribute_message(k, v)) for k, v in g.attrs.items()]
        return self._append(self._object_header(msgs)), bt_addr, heap_addr

    def close(self) -> None:
        meta_start = self._pos
        root_hdr, root_bt, root_heap = self._write_group(self.root)
        meta_end = self._pos
        if self._heap_objects:
            # one global heap collection for every variable-length string; attribute data written
            # above carries a zero collection address, patched here (the tuples are unique: length,
            # 0, index -- but they are located by parsing, not by searching for bytes)
            body = b""
            for i, raw in enumerate(self._heap_objects, start=1):
                body += struct.pack("<HH4xQ", i, 0, len(raw)) + _pad8(raw)
            size = max(4096, 16 + len(body) + 16)
            size += -size % 8
            free = size - 16 - len(body)
            coll = b"GCOL" + struct.pack("<B3xQ", 1, size) + body + struct.pack("<HH4xQ", 0, 0, free)
            coll_addr = self._append(coll.ljust(size, b"\0"))
            self._patch_heap_refs(meta_start, meta_end, coll_addr)
        eof = self._pos
        sb = SIGNATURE + struct.pack("<BBBBBBBB", 0, 0, 0, 0, 0, 8, 8, 0) + struct.pack("<HHI", _GROUP_LEAF_K, _GROUP_INTERNAL_K, 0)
        sb += struct.pack("<QQQQ", 0, _UNDEF, eof, _UNDEF)
        sb += struct.pack("<QQIIQQ", 0, root_hdr, 1, 0, root_bt, root_heap)
        assert len(sb) == _SUPERBLOCK_BYTES
        self._f.seek(0)
        self._f.write(sb)
        self._f.close()

    def _patch_heap_refs(self, lo: int, hi: int, coll_addr: int) -> None:
        """Fill the global heap address into every variable-length attribute value written in
        [lo, hi): the metadata is re-read and object headers are walked message by message."""
        self._f.flush()
        with open(self._f.name, "r+b") as f:
            f.seek(lo)
            blob = bytearray(f.read(hi - lo))
            vlen = _pad8(_vlen_string_type())
            space = _pad8(dataspace_message(()))
            pos = 0
            while True:
                pos = blob.find(vlen + space, pos)
                if pos < 0:
                    break
                at = pos + len(vlen) + len(space)
                length, addr, index = struct.unpack_from("<IQI", blob, at)
                if addr == 0 and 1 <= index <= len(self._heap_objects) and length == len(self._heap_objects[index - 1]):
                    struct.pack_into("<IQI", blob, at, length, coll_addr, index)
                pos = at
            f.seek(lo)
            f.write(blob)


# ---- the index file of the reference ----------------------------------------------------------------
def write_index(path: Path, *, dim: int, dtype, chunk_size: int, max_id_length: int, num_vectors: int,
                vector_blocks: Iterable[np.ndarray], doc_ids: Iterable, psg_ids: Iterable, ff_version: str,
                quantizer: tuple[dict, dict, dict] | None = None) -> None:
    """An ``OnDiskIndex`` file as the reference lays it out (``index/disk.py:83-85, 123-165``)."""
    capacity = -(-max(num_vectors, 1) // chunk_size) * chunk_size
    ids_dt = np.dtype(f"S{max_id_length}")

    def id_blocks(ids: Iterable) -> Iterable[np.ndarray]:
        buf = []
        for x in ids:
            raw = b"" if x is None else str(x).encode("utf-8")
            if len(raw) > max_id_length:
                raise RuntimeError(f"ID {x} is longer than the maximum ({max_id_length} characters).")
            buf.append(raw)
            if len(buf) == chunk_size:
                yield np.array(buf, dtype=ids_dt)
                buf = []
        if buf:
            yield np.array(buf, dtype=ids_dt)

    w = H5Writer(path)
    w.set_attr("/", "num_vectors", int(num_vectors))
    w.set_attr("/", "ff_version", str(ff_version))
    w.write_chunked("vectors", dtype, capacity, (dim,), chunk_size, vector_blocks)
    w.write_chunked("doc_ids", ids_dt, capacity, (), chunk_size, id_blocks(doc_ids))
    w.write_chunked("psg_ids", ids_dt, capacity, (), chunk_size, id_blocks(psg_ids))
    if quantizer is not None:
        meta, attributes, data = quantizer
        for group, values in (("quantizer/meta", meta), ("quantizer/attributes", attributes)):
            w.create_group(group)
            for k, v in values.items():
                w.set_attr(group, k, v)
        w.create_group("quantizer/data")
        for k, v in data.items():
            w.write_contiguous(f"quantizer/data/{k}", np.asarray(v))
    w.close()
