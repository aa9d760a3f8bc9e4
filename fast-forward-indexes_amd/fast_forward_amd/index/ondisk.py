"""``OnDiskIndex``: an index kept in a file and streamed through HBM while scoring.

Interface of ``fast_forward.index.OnDiskIndex`` (reference ``src/fast_forward/index/disk.py:25-418``):
same constructor / ``load`` / ``to_memory`` / ``add`` behaviour and errors.  Two things differ:

* **File format.**  The reference writes HDF5 through h5py, which is not available here, so the
  file is this build's own single-file container with the same logical content and the same
  chunking idea (the reference memory-maps raw HDF5 chunks by byte offset, disk.py:100-121):

      [0, 1 MiB)      header: magic ``FFXD0001``, uint64 length, JSON
                      {num_vectors, dim, dtype, chunk_size, max_id_length, ff_version,
                       chunks: [byte offset, ...], quantizer: {offset, length} | null}
      quantizer blob  ``Quantizer.serialize()`` as an .npz (only replaceable while the index is empty)
      chunk blocks    per chunk: vectors [chunk_size, dim] | doc_ids S<L>[chunk_size] | psg_ids S<L>[chunk_size]

  Reading HDF5 files written by the reference is future work.

* **Scoring.**  ``Index.__call__`` on this class streams only the rows a run needs: the candidate
  rows are de-duplicated and cut into windows of ``window_bytes``; an uploader thread reads window
  w+1 from the memory-mapped file and copies it to one of two HBM buffers while the HIP kernels
  score window w from the other (ctypes releases the GIL, so the copy and the kernels overlap);
  row scores are collected on the host and aggregated/interpolated/ordered once at the end by
  ``ffx_aggregate``.  Same arithmetic as the resident index, bit for bit.
"""

from __future__ import annotations

import io
import json
import logging
import queue
import struct
import os
import threading
import time
from collections.abc import Iterable, Iterator
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

from fast_forward_amd import native
from fast_forward_amd.encoder import Encoder
from fast_forward_amd.index import h5read
from fast_forward_amd.index.api import IDSequence, Index, Mode
from fast_forward_amd.index.hbm import InMemoryIndex
from fast_forward_amd.index.ids import FIRSTP, PASSAGE, IdTable
from fast_forward_amd.quantizer import Quantizer

LOGGER = logging.getLogger(__name__)

_MAGIC = b"FFXD0001"
_HEADER_BYTES = 1 << 20
_FORMAT_VERSION = 1
# gather threads of a streamed window: by default three quarters of what this process may use (cgroup quota included --
# busy threads beyond the quota get the whole group throttled, and the scoring thread and the copy engine's driver
# thread need their share: on a 16-core quota 12 threads gather at 53 GB/s, 16 at 45, 8 at 42: profiles/r03/
# ondisk_stream_threads.json); FFX_READ_THREADS overrides
def _available_host_bytes() -> int | None:
    """What the kernel says could still be used without swapping (/proc/meminfo MemAvailable), capped by what the
    container's memory limit leaves (cgroup v2 memory.max - memory.current)."""
    available = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                available = int(line.split()[1]) << 10
                break
        limit, used = open("/sys/fs/cgroup/memory.max").read().strip(), open("/sys/fs/cgroup/memory.current").read().strip()
        if limit != "max" and available is not None:
            available = min(available, max(0, int(limit) - int(used)))
    except (OSError, ValueError):
        pass
    return available


def _resident_fraction(maps: list, samples: int = 1024) -> float:
    """Share of the maps' pages that are in memory already (mincore on evenly spaced sample pages): page-locking a
    file that is NOT cached reads all of it from its disk first."""
    import ctypes

    try:
        libc = ctypes.CDLL(None, use_errno=True)
        mincore = libc.mincore
    except (OSError, AttributeError):
        return 0.0
    mincore.argtypes, mincore.restype = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p], ctypes.c_int
    page, total = 4096, sum(c.nbytes for c in maps)
    if total == 0:
        return 1.0
    seen = hit = 0
    flag = ctypes.create_string_buffer(1)
    for c in maps:
        n_here = max(1, samples * c.nbytes // total)
        lo, n_pages = c.ctypes.data // page * page, max(1, c.nbytes // page)
        for k in range(n_here):
            if mincore(lo + (k * n_pages // n_here) * page, page, flag) != 0:
                return 0.0
            seen += 1
            hit += flag.raw[0] & 1
    return hit / max(1, seen)


_READ_THREADS = int(os.environ.get("FFX_READ_THREADS", "0")) or max(1, min(24, native.usable_cores() * 3 // 4))


class OnDiskIndex(Index):
    """Fast-Forward index stored in a file; rows are streamed to the GPU on demand."""

    def __init__(self, index_file: Path, query_encoder: Encoder | None = None, quantizer: Quantizer | None = None,
                 mode: Mode = Mode.MAXP, encoder_batch_size: int = 32, init_size: int = 2**16, chunk_size: int = 2**16,
                 max_id_length: int = 8, overwrite: bool = False, memory_mapped: bool = False,
                 max_indexing_size: int = 2**10, device: int = 0, exact_sum: bool = True,
                 window_bytes: int = 1 << 30, page_lock: bool | None = None) -> None:
        """Create (or overwrite) an index file.

        :param init_size: kept for API compatibility (chunks are allocated as rows arrive).
        :param chunk_size: rows per chunk block of the file.
        :param max_id_length: maximum length (bytes) of document and passage ids.
        :param memory_mapped, max_indexing_size: kept for API compatibility; rows are always read
            through memory maps of the chunk blocks.
        :param window_bytes: HBM staging budget per window while scoring (two windows are resident).
        :param page_lock: page-lock the file's vector maps at the first streamed call, so that the device reads the rows
            it needs straight out of the mapped file (no host threads, no staging copy).  The file's pages then stay in
            host memory.  None (default): when the vectors fit in half of the memory that is available and at least
            90 % of the file is in memory already (page cache or tmpfs).
        :raises ValueError: when the file exists and ``overwrite`` is false.
        """
        index_file = Path(index_file)
        if index_file.exists() and not overwrite:
            raise ValueError(f"File {index_file} exists.")
        native.lib()
        self._index_file = index_file.absolute()
        self._init_state(device, exact_sum, window_bytes, page_lock)
        self._meta = {"version": _FORMAT_VERSION, "ff_version": "0.8.0+amd", "num_vectors": 0, "dim": None, "dtype": None,
                      "chunk_size": int(chunk_size), "max_id_length": int(max_id_length), "chunks": [], "quantizer": None}
        self._index_file.parent.mkdir(parents=True, exist_ok=True)
        with open(self._index_file, "wb") as f:
            f.truncate(_HEADER_BYTES)
        self._write_header()
        super().__init__(query_encoder=query_encoder, quantizer=quantizer, mode=mode, encoder_batch_size=encoder_batch_size)

    def _init_state(self, device: int, exact_sum: bool, window_bytes: int, page_lock: bool | None = None) -> None:
        self._page_lock, self._locked_maps, self._lock_refused = page_lock, None, None
        self._ids = IdTable()
        self._maps: list[tuple[np.memmap, np.memmap, np.memmap]] = []
        self._device, self.exact_sum, self._window_bytes = int(device), bool(exact_sum), int(window_bytes)
        self.last_stats: dict | None = None
        self._windows: tuple | None = None  # (key, device buffers, plans) kept between scoring calls
        # a SHARD VIEW of the file (load(..., shard=(rank, world))): only the rows this rank of a sharded index holds are
        # known here -- row numbers are local (0 .. len-1, file order), _row_map[local] is the row of the file
        self._row_map: np.ndarray | None = None

    def _release_windows(self) -> None:
        if getattr(self, "_windows", None):
            _, bufs, plans = self._windows
            for p in plans:
                p.close()
            for b in bufs:
                b.close()
        self._windows = None

    def _unlock_maps(self) -> None:
        if getattr(self, "_locked_maps", None) is not None:
            self._release_windows()  # (their chunk tables name the maps' device addresses)
            self._locked_maps.close()
        self._locked_maps = None

    def _device_readable_maps(self, chunk_maps: list) -> bool:
        """Page-lock the vector maps (once per set of maps) if the policy allows: True when the device can pull rows out
        of them itself.  A refusal (policy, or the driver's) is remembered in ``last_stats['row_source']``."""
        if self._locked_maps is not None:
            return True
        if self._page_lock is False or self._lock_refused is not None:
            return False
        total = sum(c.nbytes for c in chunk_maps)
        if self._page_lock is None:
            available = _available_host_bytes()
            if available is None or total > available // 2:
                self._lock_refused = f"host gather threads (vector maps of {total >> 20} MiB not page-locked: more than half of the available host memory)"
                return False
            cached = _resident_fraction(chunk_maps)
            if cached < 0.9:  # page-locking would read the rest of the file from its disk first
                self._lock_refused = f"host gather threads (vector maps not page-locked: {cached:.0%} of the file is in memory)"
                return False
        try:
            self._locked_maps = native.PinnedMaps(chunk_maps)
        except (RuntimeError, MemoryError) as e:
            if self._page_lock:
                raise
            self._lock_refused = f"host gather threads (page-locking the vector maps failed: {e})"
            return False
        return True

    def __del__(self):
        try:
            self._release_windows()
            self._unlock_maps()
        except Exception:  # interpreter shutdown
            pass

    # ---- file plumbing ----------------------------------------------------------------------------
    def _write_header(self) -> None:
        blob = json.dumps(self._meta).encode()
        if len(blob) + 16 > _HEADER_BYTES:
            raise RuntimeError("index header overflow: use a larger chunk_size")
        with open(self._index_file, "r+b") as f:
            f.write(_MAGIC + struct.pack("<Q", len(blob)) + blob)

    @staticmethod
    def _read_header(path: Path) -> dict:
        with open(path, "rb") as f:
            head = f.read(16)
            if head[:8] != _MAGIC:
                raise ValueError(f"{path} is not an index file of this library (HDF5 files of the reference are not readable yet)")
            return json.loads(f.read(struct.unpack("<Q", head[8:])[0]))

    def _chunk_layout(self) -> tuple[int, int, int]:
        m = self._meta
        vec_bytes = m["chunk_size"] * m["dim"] * np.dtype(m["dtype"]).itemsize
        id_bytes = m["chunk_size"] * m["max_id_length"]
        return vec_bytes, id_bytes, vec_bytes + 2 * id_bytes

    def _open_maps(self) -> None:
        """(Re)create one memory map triple per chunk block (the reference's per-chunk np.memmap)."""
        m = self._meta
        self._unlock_maps()  # before the old maps go away
        self._lock_refused = None
        self._maps = []
        if not m["chunks"]:
            return
        vec_bytes, id_bytes, _ = self._chunk_layout()
        ids_dt = np.dtype(f"S{m['max_id_length']}")
        for off in m["chunks"]:
            vec = np.memmap(self._index_file, mode="r+", dtype=np.dtype(m["dtype"]), offset=off, shape=(m["chunk_size"], m["dim"]))
            docs = np.memmap(self._index_file, mode="r+", dtype=ids_dt, offset=off + vec_bytes, shape=(m["chunk_size"],))
            psgs = np.memmap(self._index_file, mode="r+", dtype=ids_dt, offset=off + vec_bytes + id_bytes, shape=(m["chunk_size"],))
            self._maps.append((vec, docs, psgs))

    def _ensure_chunks(self, n_total: int) -> None:
        m = self._meta
        need = -(-n_total // m["chunk_size"])
        if need <= len(m["chunks"]):
            return
        _, _, block = self._chunk_layout()
        with open(self._index_file, "r+b") as f:
            f.seek(0, 2)
            end = max(f.tell(), _HEADER_BYTES)
            while len(m["chunks"]) < need:
                m["chunks"].append(end)
                end += block
            f.truncate(end)  # sparse until written
        self._open_maps()

    def _on_quantizer_set(self) -> None:
        """Store the serialized quantizer in the file (disk.py:123-136); only possible while empty."""
        meta, attributes, data = self._quantizer.serialize()
        buf = io.BytesIO()
        np.savez(buf, __meta__=np.frombuffer(json.dumps(dict(meta)).encode(), dtype=np.uint8),
                 __attributes__=np.frombuffer(json.dumps(dict(attributes)).encode(), dtype=np.uint8), **data)
        blob = buf.getvalue()
        with open(self._index_file, "r+b") as f:
            f.truncate(_HEADER_BYTES)
            f.seek(_HEADER_BYTES)
            f.write(blob)
        self._meta["quantizer"] = {"offset": _HEADER_BYTES, "length": len(blob)}
        self._write_header()

    @staticmethod
    def _load_quantizer(path: Path, where: dict) -> Quantizer:
        with open(path, "rb") as f:
            f.seek(where["offset"])
            z = np.load(io.BytesIO(f.read(where["length"])))
        meta = json.loads(bytes(z["__meta__"]).decode())
        attributes = json.loads(bytes(z["__attributes__"]).decode())
        data = {k: z[k] for k in z.files if not k.startswith("__")}
        return Quantizer.deserialize(meta, attributes, data)

    # ---- Index storage interface ------------------------------------------------------------------------
    def _get_num_vectors(self) -> int:
        return int(self._meta["num_vectors"]) if self._row_map is None else int(self._row_map.shape[0])

    def _file_rows(self, rows: np.ndarray) -> np.ndarray:
        """Row numbers of the FILE for this index's row numbers (the same, unless this is a shard view)."""
        rows = np.asarray(rows, dtype=np.int64)
        return rows if self._row_map is None else self._row_map[rows]

    def _get_internal_dim(self) -> int | None:
        return self._meta["dim"]

    def _get_doc_ids(self) -> set[str]:
        return self._ids.doc_ids

    def _get_psg_ids(self) -> set[str]:
        return self._ids.psg_ids

    def _add(self, vectors: np.ndarray, doc_ids: IDSequence, psg_ids: IDSequence) -> None:
        m = self._meta
        if self._row_map is not None:
            raise RuntimeError("a shard view of an index file is read-only")
        if m.get("read_only"):
            raise RuntimeError("HDF5 index files of the reference are opened read-only; use to_memory() or copy into a new OnDiskIndex")
        limit = m["max_id_length"]
        # every check first: the index must stay unchanged if anything is wrong (disk.py:207-241)
        for d in doc_ids:
            if d is not None and len(d.encode()) > limit:
                raise RuntimeError(f"Document ID {d} is longer than the maximum ({limit} characters).")
        for p in psg_ids:
            if p is not None and len(p.encode()) > limit:
                raise RuntimeError(f"Passage ID {p} is longer than the maximum ({limit} characters).")
        if m["dim"] is None:
            if self._quantizer is not None:
                dtype = np.dtype(vectors.dtype)  # codes
            else:
                dtype = np.dtype(np.float16) if vectors.dtype == np.float16 else np.dtype(np.float32)
            m["dim"], m["dtype"] = int(vectors.shape[1]), dtype.name
        n0, n_new = m["num_vectors"], vectors.shape[0]
        self._ids.add(doc_ids, psg_ids, n0)  # duplicate passage id -> RuntimeError, nothing modified
        self._ensure_chunks(n0 + n_new)
        data = np.ascontiguousarray(vectors, dtype=np.dtype(m["dtype"]))
        docs = np.array([(d or "").encode() for d in doc_ids], dtype=f"S{limit}")
        psgs = np.array([(p or "").encode() for p in psg_ids], dtype=f"S{limit}")
        cs, done = m["chunk_size"], 0
        while done < n_new:
            c, lo = divmod(n0 + done, cs)
            take = min(n_new - done, cs - lo)
            vec, dmap, pmap = self._maps[c]
            vec[lo:lo + take] = data[done:done + take]
            dmap[lo:lo + take] = docs[done:done + take]
            pmap[lo:lo + take] = psgs[done:done + take]
            done += take
        for maps in self._maps:
            for mm in maps:
                mm.flush()
        m["num_vectors"] = n0 + n_new
        self._write_header()

    def _read_rows(self, rows: np.ndarray) -> np.ndarray:
        """Stored rows (vectors or codes) for global row numbers, in the order given.  Large reads
        are split over a few threads: the gather out of the page cache is a plain memory copy
        (NumPy releases the GIL for it) and one core moves only ~3 GB/s."""
        rows = self._file_rows(rows)
        out = np.empty((rows.shape[0], self._meta["dim"]), dtype=np.dtype(self._meta["dtype"]))
        cs = self._meta["chunk_size"]

        def gather(lo: int, hi: int) -> None:
            chunk, local = np.divmod(rows[lo:hi], cs)
            for c in np.unique(chunk):
                sel = np.flatnonzero(chunk == c)
                out[lo + sel] = self._maps[int(c)][0][local[sel]]

        n = rows.shape[0]
        n_threads = min(_READ_THREADS, max(1, n * out.shape[1] * out.dtype.itemsize >> 24))  # >= 16 MiB per thread
        if n_threads <= 1:
            gather(0, n)
        else:
            bounds = np.linspace(0, n, n_threads + 1, dtype=np.int64)
            with ThreadPoolExecutor(n_threads) as pool:
                list(pool.map(lambda i: gather(int(bounds[i]), int(bounds[i + 1])), range(n_threads)))
        return out

    def _get_vectors(self, ids: Iterable[str]) -> tuple[np.ndarray, list[str]]:
        rows, owners = self._ids.rows_for(ids, self.mode.value)
        if not rows:
            return np.array([]), []
        return self._read_rows(np.asarray(rows)), owners

    def _batch_iter(self, batch_size: int) -> Iterator[tuple[np.ndarray, IDSequence, IDSequence]]:
        n = len(self)
        docs, psgs = self._ids.row_ids(n)
        for lo in range(0, n, batch_size):
            hi = min(lo + batch_size, n)
            yield self._read_rows(np.arange(lo, hi)), docs[lo:hi], psgs[lo:hi]

    def _resolve_ids(self, ids: np.ndarray, mode: Mode) -> np.ndarray:
        return self._ids.resolve(ids, mode.value)

    def _native_id_table(self, mode: Mode):
        return self._ids.native_table(mode.value)

    # ---- persistence ----------------------------------------------------------------------------------------
    @classmethod
    def load(cls, index_file: Path, query_encoder: Encoder | None = None, mode: Mode = Mode.MAXP, encoder_batch_size: int = 32,
             memory_mapped: bool = False, max_indexing_size: int = 2**10, device: int = 0, exact_sum: bool = True,
             window_bytes: int = 1 << 30, shard: "tuple[int, int] | None" = None, page_lock: bool | None = None) -> "OnDiskIndex":
        """Open an existing index file (disk.py:355-418): ids are read back into the host tables,
        the stored quantizer (if any) is reconstructed.

        ``shard=(rank, world)``: a SHARD VIEW for one rank of a sharded index -- only the rows whose document id
        (passage id, for rows without one) hashes to ``rank`` (crc32 % world, ``sharded._owner_of``) are kept: their ids
        are the only strings interned (host memory ~ N / world per rank, where the reference -- one process -- reads
        all N, disk.py:355-418), row numbers are local, read-only."""
        index_file = Path(index_file)
        native.lib()
        index = cls.__new__(cls)
        index._index_file = index_file.absolute()
        index._init_state(device, exact_sum, window_bytes, page_lock)
        if shard is not None and not (0 <= int(shard[0]) < int(shard[1]) <= 254):
            raise ValueError("shard must be (rank, world) with 0 <= rank < world <= 254")
        if h5read.is_hdf5(index_file):  # an index written by the reference (h5py): opened read-only
            index._load_hdf5(query_encoder, mode, encoder_batch_size, shard)
            return index
        index._meta = cls._read_header(index_file)
        Index.__init__(index, query_encoder=query_encoder, quantizer=None, mode=mode, encoder_batch_size=encoder_batch_size)
        if index._meta.get("quantizer"):
            index._quantizer = cls._load_quantizer(index_file, index._meta["quantizer"])
            index._quantizer._attached = True
        index._open_maps()
        n, cs = index._meta["num_vectors"], index._meta["chunk_size"]
        kept: list[np.ndarray] = []
        for c, (_, dmap, pmap) in enumerate(index._maps):
            lo = c * cs
            take = min(cs, n - lo)
            if take <= 0:
                break
            kept.append(index._register_ids(np.asarray(dmap[:take]), np.asarray(pmap[:take]), lo, shard))
        if shard is not None:
            index._row_map = np.concatenate(kept) if kept else np.zeros(0, dtype=np.int64)
        return index

    def _register_ids(self, docs: np.ndarray, psgs: np.ndarray, first_file_row: int, shard) -> np.ndarray:
        """One block of the file's id columns (fixed-width bytes, "" = no id) into the id tables; with ``shard`` only the
        rows this rank holds (``ffx_host_owner_crc32``: no Python string is made of anybody else's ids).  Returns the
        file rows that were registered."""
        rows = np.arange(first_file_row, first_file_row + docs.shape[0], dtype=np.int64)
        if shard is not None:
            mine = np.flatnonzero(native.owner_of_rows(docs, psgs, int(shard[1])) == int(shard[0]))
            docs, psgs, rows = docs[mine], psgs[mine], rows[mine]
        self._ids.add([d.decode() or None for d in docs], [p.decode() or None for p in psgs], self._ids._n_rows)
        return rows

    def _load_hdf5(self, query_encoder, mode, encoder_batch_size, shard=None) -> None:
        """Open an HDF5 index file of the reference (layout: disk.py:83-85, :123-165).  The chunk
        B-tree gives the byte offset of every ``vectors`` chunk, which is memory-mapped exactly as
        the reference's own mmap mode does (disk.py:100-121); ids are read once into the host
        tables.  The file is not modified: ``add`` raises."""
        root = h5read.read(self._index_file)
        n = int(root.attrs.get("num_vectors", 0))
        self._meta = {"version": "hdf5", "ff_version": str(root.attrs.get("ff_version", "")), "num_vectors": n, "dim": None,
                      "dtype": None, "chunk_size": 1, "max_id_length": 0, "chunks": [], "quantizer": None, "read_only": True}
        Index.__init__(self, query_encoder=query_encoder, quantizer=None, mode=mode, encoder_batch_size=encoder_batch_size)
        if "quantizer" in root.groups:
            q = root.groups["quantizer"]
            meta = dict(q.groups["meta"].attrs)
            # the reference's quantizer classes live in fast_forward.quantizer.*; same names here
            meta["__module__"] = "fast_forward_amd.quantizer"
            meta["_trained"] = bool(meta.get("_trained", False))
            attributes = {k: (bool(v) if k == "verbose" else v) for k, v in q.groups["attributes"].attrs.items()}
            data = {k: d.read(self._index_file) for k, d in q.groups["data"].datasets.items()} if "data" in q.groups else {}
            self._quantizer = Quantizer.deserialize(meta, attributes, data)
            self._quantizer._attached = True
        if "vectors" not in root.datasets:
            return
        vec = root.datasets["vectors"]
        if vec.chunk_shape is None or vec.chunk_shape[1] != vec.shape[1]:
            raise RuntimeError("This index does not support memory maps.")  # same condition as disk.py:103-107
        cs, dim = vec.chunk_shape
        self._meta.update(dim=int(dim), dtype=vec.dtype.name, chunk_size=int(cs),
                          max_id_length=int(root.datasets["doc_ids"].dtype.itemsize))
        n_chunks = -(-vec.shape[0] // cs)
        empty = np.zeros((cs, dim), dtype=vec.dtype)
        self._maps = []
        for c in range(n_chunks):
            where = vec.chunks.get((c * cs, 0))
            rows = empty if where is None else np.memmap(self._index_file, mode="r", dtype=vec.dtype, offset=where[0], shape=(cs, dim))
            self._maps.append((rows, None, None))
        # the id columns block by block (a shard view keeps ~1 / world of every block: its host memory never holds the columns
        # of the whole file, as the container format's loader above)
        d_ids, p_ids = root.datasets["doc_ids"], root.datasets["psg_ids"]
        kept = [self._register_ids(d_ids.read_rows(self._index_file, lo, min(n, lo + (1 << 20))), p_ids.read_rows(self._index_file, lo, min(n, lo + (1 << 20))),
                                   lo, shard) for lo in range(0, n, 1 << 20)]
        if shard is not None:
            self._row_map = np.concatenate(kept) if kept else np.zeros(0, dtype=np.int64)

    def to_memory(self, batch_size: int | None = None) -> InMemoryIndex:
        """The whole index resident in HBM (disk.py:177-205)."""
        index = InMemoryIndex(query_encoder=self._query_encoder, quantizer=None, mode=self.mode,
                              encoder_batch_size=self._encoder_batch_size, init_size=max(1, len(self)),
                              device=self._device, exact_sum=self.exact_sum)
        if self._quantizer is not None:
            index._quantizer = self._quantizer  # already attached and trained; rows below are codes
        n = len(self)
        step = batch_size or max(n, 1)
        docs, psgs = self._ids.row_ids(n)
        chunk_maps = [m[0] for m in self._maps]
        stored = np.dtype(self._meta["dtype"]) if self._meta["dtype"] else None
        book = self._quantizer.device_codebook() if self._quantizer is not None else None
        # stored rows can go to HBM as they are (fp32 / fp16 vectors, or codes the kernels read)
        direct = (n > 0 and stored in (np.dtype(np.float32), np.dtype(np.float16), np.dtype(np.uint8))
                  and (self._quantizer is None) == (stored != np.dtype(np.uint8)) and (self._quantizer is None or book is not None)
                  and all(isinstance(c, np.ndarray) and c.flags["C_CONTIGUOUS"] for c in chunk_maps))
        for lo in range(0, n, step):
            hi = min(lo + step, n)
            if direct:
                from types import SimpleNamespace

                shape = SimpleNamespace(shape=(hi - lo, self._meta["dim"]), dtype=stored)
                index._add(shape, docs[lo:hi], psgs[lo:hi],
                           _uploader=lambda store, at, lo=lo, hi=hi: store.upload_rows(chunk_maps, self._file_rows(np.arange(lo, hi)), at, _READ_THREADS))
            else:
                index._add(self._read_rows(np.arange(lo, hi)), docs[lo:hi], psgs[lo:hi])
        return index

    # ---- scoring: stream the needed rows through two HBM windows --------------------------------------------
    def _expand(self, pair_off: np.ndarray, pair_ids: np.ndarray, mode: int, distinct: bool = False):
        """Pairs -> gathered rows on the host (get_indices, index/util.py:29-41): (row numbers [R], segment offsets
        [P+1]); the offsets are None when every candidate contributes exactly one row (rows and pairs coincide).
        ``distinct=True``: where one native pass can do both, the first item is the pair (distinct rows ascending, slot
        of every gathered row among them) instead of the row numbers."""
        n_pairs = int(pair_off[-1])
        # the id tables as int64 arrays, made once per state of the index (a copy of a table of millions of rows per
        # call cost more than the candidates' own look-ups)
        cache = getattr(self, "_expand_tables", None)
        if cache is None or cache[0] != self._ids.version:
            doc_off, doc_rows = self._ids.doc_csr()
            doc_off = np.ascontiguousarray(doc_off, dtype=np.int64)
            doc_rows, psg_rows = np.ascontiguousarray(doc_rows, dtype=np.int64), np.ascontiguousarray(self._ids.psg_rows(), dtype=np.int64)
            small = len(self) < 2**31 - 1  # 4-byte copies for the threaded gathers (FirstP, PASSAGE, one row per document)
            tables32 = (doc_off.astype(np.int32), doc_rows.astype(np.int32), psg_rows.astype(np.int32)) if small else None
            cache = (self._ids.version, doc_off, doc_rows, psg_rows, bool(doc_off.shape[0] > 1 and (np.diff(doc_off) == 1).all()), tables32)
            self._expand_tables = cache
        _, doc_off, doc_rows, psg_rows, one_row_each, tables32 = cache
        ids32 = np.ascontiguousarray(pair_ids, dtype=np.int32)
        fused = distinct and tables32 is not None and n_pairs >= (1 << 16)
        if mode == PASSAGE:
            if fused:
                return native.rows_unique(ids32, tables32[2], None, len(self)), None
            if tables32 is not None:  # (the gathers by host threads: a million random reads of a table of millions)
                return native.take32(tables32[2], ids32), None
            return psg_rows[pair_ids], None
        if (mode == FIRSTP or one_row_each) and tables32 is not None:
            if fused:
                return native.rows_unique(ids32, tables32[0][:-1], tables32[1], len(self)), None
            return native.take32(tables32[1], native.take32(tables32[0], ids32)), None
        if fused:  # every row of every candidate's document: offsets, rows, distinct rows and slots natively
            seg_off, uniq, inverse = native.segment_rows_unique(ids32, tables32[0], tables32[1], len(self))
            return (uniq, inverse), seg_off
        start = doc_off[pair_ids]
        if mode == FIRSTP or one_row_each:
            return doc_rows[start], None
        cnt = doc_off[pair_ids.astype(np.int64) + 1] - start
        seg_off = np.zeros(n_pairs + 1, dtype=np.int64)
        np.cumsum(cnt, out=seg_off[1:])
        within = np.arange(int(seg_off[-1]), dtype=np.int64) - np.repeat(seg_off[:-1], cnt)
        return doc_rows[np.repeat(start, cnt) + within], seg_off

    def _score_pairs(self, query_vectors, pair_off, pair_ids, mode: Mode, sparse=None, alpha=None, sort=False, want_dense=True):
        if len(self) == 0:
            raise IndexError("the index is empty")
        q = np.ascontiguousarray(query_vectors, dtype=np.float32)
        book = self._quantizer.device_codebook() if self._quantizer is not None else None
        host_decode = self._quantizer is not None and book is None
        dim_q = self._quantizer.dims[0] if self._quantizer is not None else self._meta["dim"]
        if q.ndim != 2 or q.shape[1] != dim_q:
            raise ValueError(f"Query vector dimensionality ({q.shape[-1]}) does not match index dimensionality ({dim_q}).")
        if book is not None and book[1] is not None:
            q = np.ascontiguousarray(q @ book[1], dtype=np.float32)  # OPQ: rotate the queries (see hbm.py)
        n_q = q.shape[0]
        pair_off = np.asarray(pair_off, dtype=np.int64)
        t_call = time.perf_counter()
        rows, seg_off = self._expand(pair_off, np.asarray(pair_ids), mode.value, distinct=True)
        row_off = pair_off if seg_off is None else seg_off[pair_off]  # first gathered row of every query
        t_expanded = time.perf_counter()
        # each needed row is read once: the distinct rows, ascending, and every gathered row's slot among them
        if isinstance(rows, tuple):  # (one native pass did the expansion and this)
            uniq, slot_row = rows
        elif rows.shape[0] >= (1 << 16):
            uniq, slot_row = native.unique_rows(rows, len(self))
        else:
            uniq, slot_row = np.unique(rows, return_inverse=True)
        n_rows_gathered = int(slot_row.shape[0])
        store_dtype = np.dtype(self._meta["dtype"])
        if host_decode or store_dtype not in (np.dtype(np.float16), np.dtype(np.uint8)):
            store_dtype = np.dtype(np.float32)  # float64 files of the reference are scored in fp32 (divergences)
        store_dim = dim_q if host_decode else self._meta["dim"]
        per_window = max(1, self._window_bytes // max(1, store_dim * store_dtype.itemsize))
        n_win = -(-uniq.shape[0] // per_window)
        rs = np.empty(n_rows_gathered, dtype=np.float32)
        t_begin = time.perf_counter()
        stats = {"s_upload": 0.0, "s_score": 0.0, "windows": n_win, "unique_rows": int(uniq.shape[0]), "gathered_rows": n_rows_gathered, "ms_gather_dot": 0.0,
                 "bytes_streamed": int(uniq.shape[0]) * store_dim * store_dtype.itemsize}

        # the two HBM windows (and their plans, page-locked staging included) live on between calls
        key = (store_dim, store_dtype.str, per_window, id(self._quantizer))
        if self._windows is not None and self._windows[0] != key:
            self._release_windows()
        if self._windows is None:
            self._windows = (key, [], [])
        _, bufs, plans = self._windows
        while len(bufs) < min(2, n_win):
            buf = native.DeviceIndex(per_window, store_dim, self._device, dtype=store_dtype)
            if book is not None:
                buf.set_codebook(book[0])
            buf.set_psg_rows(np.arange(buf.capacity, dtype=np.int32))
            bufs.append(buf)
            plans.append(native.Plan(buf))
        bufs, plans = bufs[:min(2, n_win)], plans[:min(2, n_win)]
        for plan in plans:
            plan.set_queries(q)
        chunk_maps = [m[0] for m in self._maps]
        native_gather = (not host_decode and np.dtype(self._meta["dtype"]) == store_dtype
                         and all(isinstance(c, np.ndarray) and c.flags["C_CONTIGUOUS"] for c in chunk_maps))
        device_pull = native_gather and self._device_readable_maps(chunk_maps)
        stats["row_source"] = ("device pull out of the page-locked file maps" if device_pull else self._lock_refused or
                               ("host gather threads" if native_gather else "host reads (NumPy)"))
        free, ready = queue.Queue(), queue.Queue()
        for i in range(len(bufs)):
            free.put(i)
        failure: list[BaseException] = []
        stop = threading.Event()

        def uploader() -> None:
            last_pull = None
            try:
                for w in range(n_win):
                    i = free.get()
                    if i is None or stop.is_set():
                        return
                    want = uniq[w * per_window:(w + 1) * per_window]
                    t_up = time.perf_counter()
                    if device_pull:  # the device reads the rows out of the page-locked maps itself, over PCIe; one window
                        # after the other (sharing the link they would both arrive late, and the link would idle while both are scored)
                        if last_pull is not None and last_pull != i:
                            bufs[i].upload_after(bufs[last_pull])
                        bufs[i].pull_rows(chunk_maps, self._file_rows(want), 0)
                        last_pull = i
                    elif native_gather:  # threads of libffx: chunk maps -> page-locked staging -> HBM, overlapped; the last
                        # copies are still in flight when this returns (the scorer waits for them), so the next window's
                        # gather starts right away
                        bufs[i].upload_rows(chunk_maps, self._file_rows(want), 0, _READ_THREADS, wait=False)
                    else:
                        data = self._read_rows(want)
                        if host_decode:
                            data = np.ascontiguousarray(self._quantizer.decode(data), dtype=np.float32)
                        bufs[i].upload(data, 0)
                    stats["s_upload"] += time.perf_counter() - t_up
                    ready.put((w, i))
            except BaseException as e:  # surfaced in the scoring thread
                failure.append(e)
                ready.put((-1, -1))

        t = threading.Thread(target=uploader, name="ffx-ondisk-uploader", daemon=True)
        t.start()
        try:
            # a window's gathered rows: their positions in the row list (ascending, so grouped by query), the queries'
            # offsets among them and the rows' places in the window.  Made while the window is still on its way.
            def candidates_of(w: int):
                slots = np.flatnonzero((slot_row >= w * per_window) & (slot_row < (w + 1) * per_window))
                return slots, np.searchsorted(slots, row_off).astype(np.int64), (slot_row[slots] - w * per_window).astype(np.int32)

            upcoming = candidates_of(0)
            for _ in range(n_win):
                w, i = ready.get()
                if w < 0:
                    raise failure[0]
                t_sc = time.perf_counter()
                if native_gather:
                    bufs[i].upload_wait()
                slots, local_off, local_rows = upcoming
                plans[i].set_pairs(local_off, local_rows, None, PASSAGE)
                plans[i].run(0.0, exact_sum=self.exact_sum)
                if w + 1 < n_win:
                    upcoming = candidates_of(w + 1)  # (windows come in order: one uploader)
                rs[slots] = plans[i].fetch()[0]
                stats["ms_gather_dot"] += plans[i].stats()["ms_gather_dot"]
                stats["kernel_variant"] = plans[i].stats()["kernel_variant"]
                stats["s_score"] += time.perf_counter() - t_sc
                free.put(i)
        finally:
            stop.set()
            free.put(None)  # unblock the uploader if the scoring thread failed first
            t.join()
            if device_pull:  # after a failure a pull may still be reading the maps: nothing outlives the call
                for b in bufs:
                    try:
                        b.upload_wait()
                    except native.FFXError:
                        pass
        stats["s_windows"] = time.perf_counter() - t_begin
        stats["s_expand"], stats["s_unique_rows"] = t_expanded - t_call, t_begin - t_expanded
        t_agg = time.perf_counter()
        out = native.aggregate(rs, seg_off, mode.value, sparse, 0.0 if alpha is None else alpha, pair_off, sort=sort, device=self._device)
        stats["s_aggregate"] = time.perf_counter() - t_agg
        self.last_stats = stats
        return out
