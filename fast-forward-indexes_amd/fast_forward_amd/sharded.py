"""Row-sharded index over the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference is single-process and has no counterpart; this is the multi-GPU form of the same
path.  The embedding matrix is sharded by *document* (a document's
passages never straddle ranks, so MaxP/FirstP/AveP stay rank-local), query vectors are
replicated, and every rank scores the candidates whose documents it holds.  The one exchange
step is an all-to-all of dense scores: query ``q`` is *owned* by one rank, which receives q's
scores from every rank, puts them back in first-stage order, interpolates and orders them.
Per step each rank sends/receives ~P/world * 4 bytes, spread over all peers at once -- with 7
point-to-point xGMI links per GPU a full-mesh all-to-all uses every link, where a ring
all-gather of all scores would be bound by one.

Everything on the data path is native and behind the C ABI (``include/ffx.h``: ``ffx_comm_*`` =
communicator over RCCL or shared memory, ``ffx_route`` = routing, ``ffx_sharded_*`` = local plans +
exchange + owner merge); this module is the ctypes-thin Python face of it plus the process
bootstrap (handing the communicator id from rank 0 to the others through a file).  No PyTorch.

``route`` below is the NumPy statement of ``ffx_route``: the CPU tests check the native routing
against it, and the world_size-2 gloo tests drive it with CPU tensors.
"""

from __future__ import annotations

import os
import tempfile
import time
from dataclasses import dataclass

import numpy as np


# ---- partitioning ------------------------------------------------------------------------------
def doc_partition(doc_off: np.ndarray, world: int) -> np.ndarray:
    """Document boundaries ``b[0..world]`` of contiguous shards with ~equal row counts; shard r
    holds documents ``b[r]..b[r+1]`` = rows ``doc_off[b[r]]..doc_off[b[r+1]]``."""
    doc_off = np.asarray(doc_off, dtype=np.int64)
    n_docs, n_rows = doc_off.shape[0] - 1, int(doc_off[-1])
    targets = (np.arange(world + 1, dtype=np.int64) * n_rows) // world
    bounds = np.searchsorted(doc_off, targets, side="left").astype(np.int64)
    bounds[0], bounds[-1] = 0, n_docs
    return np.maximum.accumulate(bounds)


def query_partition(n_queries: int, world: int) -> np.ndarray:
    """Owner rank r finalises queries ``b[r]..b[r+1]`` (contiguous blocks of ~equal size)."""
    return (np.arange(world + 1, dtype=np.int64) * n_queries) // world


@dataclass
class Route:
    """Everything rank ``rank`` needs for one run (all host arrays)."""

    rank: int
    world: int
    # scoring side: the candidates whose documents this rank holds, as a local CSR over ALL queries
    local_pair_off: np.ndarray  # int64[Q+1]
    local_pair_ids: np.ndarray  # int32[P_r]  ordinals relative to the shard
    send_counts: np.ndarray     # int64[world]: local scores going to each owner (contiguous slices)
    # owner side: the queries this rank finalises
    q_lo: int
    q_hi: int
    own_pair_off: np.ndarray    # int64[Qo+1]: their candidates in first-stage order
    recv_counts: np.ndarray     # int64[world]: scores arriving from each rank
    recv_slot: np.ndarray       # int32[P_o]: position of candidate p in the received buffer
    own_sparse: np.ndarray | None
    own_first_pair: int         # global position of the first owned pair


def route(pair_off, pair_ids, doc_bounds, rank: int, world: int, sparse=None) -> Route:
    pair_off = np.asarray(pair_off, dtype=np.int64)
    pair_ids = np.asarray(pair_ids, dtype=np.int64)
    doc_bounds = np.asarray(doc_bounds, dtype=np.int64)
    n_q = pair_off.shape[0] - 1
    holder = (np.searchsorted(doc_bounds, pair_ids, side="right") - 1).astype(np.int32)
    if pair_ids.size and (pair_ids.min() < 0 or pair_ids.max() >= doc_bounds[-1]):
        bad = int(np.flatnonzero((pair_ids < 0) | (pair_ids >= doc_bounds[-1]))[0])
        raise IndexError(f"pair {bad} names document ordinal {int(pair_ids[bad])}, outside the sharded index")
    pair_q = np.repeat(np.arange(n_q, dtype=np.int64), np.diff(pair_off))
    mine = holder == rank
    local_ids = (pair_ids[mine] - doc_bounds[rank]).astype(np.int32)
    local_off = np.zeros(n_q + 1, dtype=np.int64)
    np.cumsum(np.bincount(pair_q[mine], minlength=n_q), out=local_off[1:])
    q_bounds = query_partition(n_q, world)
    send_counts = np.diff(local_off[q_bounds])
    q_lo, q_hi = int(q_bounds[rank]), int(q_bounds[rank + 1])
    lo, hi = int(pair_off[q_lo]), int(pair_off[q_hi])
    own_holder = holder[lo:hi]
    recv_counts = np.bincount(own_holder, minlength=world).astype(np.int64)
    arrival = np.argsort(own_holder, kind="stable")  # arrival order: by source rank, then first-stage order
    recv_slot = np.empty(hi - lo, dtype=np.int32)
    recv_slot[arrival] = np.arange(hi - lo, dtype=np.int32)
    return Route(
        rank=rank, world=world, local_pair_off=local_off, local_pair_ids=local_ids, send_counts=send_counts,
        q_lo=q_lo, q_hi=q_hi, own_pair_off=pair_off[q_lo:q_hi + 1] - lo, recv_counts=recv_counts, recv_slot=recv_slot,
        own_sparse=None if sparse is None else np.ascontiguousarray(sparse[lo:hi], dtype=np.float32), own_first_pair=lo,
    )


# ---- owner-side routing: every rank is handed only the queries it owns ---------------------------------------------
@dataclass
class OwnedRoute:
    """What rank ``rank`` derives from ITS block of a batch (the NumPy statement of ``ffx_sharded_stage_owned``'s device
    kernels): two messages per holder and the owner-side arrival slots."""

    rank: int
    world: int
    counts: np.ndarray    # int32[world, Qo]: counts[h, q] = candidates of owned query q that rank h holds (message A to h: counts[h])
    starts: np.ndarray    # int64[world + 1]: the run of ordinals for holder h is send_ids[starts[h]:starts[h + 1]] (message B)
    send_ids: np.ndarray  # int32[Po]: holder-local ordinals grouped by holder, queries ascending, first-stage order inside
    slot: np.ndarray      # int32[Po]: where the score of owned candidate p arrives (what the owner-side merge reads)


def route_owned(own_pair_off, own_pair_ids, rank: int, world: int, doc_bounds=None, holder=None) -> OwnedRoute:
    """Owner side of a fresh batch.  ``own_pair_ids``: global document ordinals (holders by ``doc_bounds``) or, with
    ``holder``, the ordinal on rank ``holder[p]``."""
    off = np.asarray(own_pair_off, dtype=np.int64)
    ids = np.asarray(own_pair_ids, dtype=np.int64)
    n_q = off.shape[0] - 1
    if holder is None:
        bounds = np.asarray(doc_bounds, dtype=np.int64)
        if ids.size and (ids.min() < bounds[0] or ids.max() >= bounds[-1]):
            bad = int(np.flatnonzero((ids < bounds[0]) | (ids >= bounds[-1]))[0])
            raise IndexError(f"pair {bad} names document ordinal {int(ids[bad])}, outside the sharded index")
        who = (np.searchsorted(bounds, ids, side="right") - 1).astype(np.int64)
        local = ids - bounds[who]
    else:
        who = np.asarray(holder, dtype=np.int64)
        local = ids
    pair_q = np.repeat(np.arange(n_q, dtype=np.int64), np.diff(off))
    counts = np.zeros((world, n_q), dtype=np.int32)
    np.add.at(counts, (who, pair_q), 1)
    arrival = np.lexsort((np.arange(ids.shape[0]), pair_q, who))  # by holder, then query, then first-stage position
    slot = np.empty(ids.shape[0], dtype=np.int32)
    slot[arrival] = np.arange(ids.shape[0], dtype=np.int32)
    starts = np.concatenate([[0], np.cumsum(counts.sum(axis=1, dtype=np.int64))]).astype(np.int64)
    return OwnedRoute(rank=rank, world=world, counts=counts, starts=starts, send_ids=local[arrival].astype(np.int32), slot=slot)


def assemble_local(q_bounds, counts_from, ids_from):
    """Holder side: the messages of the owners, in rank order, ARE this rank's local CSR.  ``counts_from[o]``: int32[Q_o]
    per-query counts from owner o, ``ids_from[o]``: its ordinals.  Returns (local_pair_off, local_pair_ids, send_counts:
    how many local scores go back to each owner)."""
    q_bounds = np.asarray(q_bounds, dtype=np.int64)
    per_q = np.concatenate([np.asarray(c, dtype=np.int64) for c in counts_from]) if len(counts_from) else np.zeros(0, dtype=np.int64)
    assert per_q.shape[0] == int(q_bounds[-1])
    local_off = np.concatenate([[0], np.cumsum(per_q)]).astype(np.int64)
    local_ids = np.concatenate([np.asarray(i, dtype=np.int32) for i in ids_from]) if len(ids_from) else np.zeros(0, dtype=np.int32)
    return local_off, local_ids.astype(np.int32), np.diff(local_off[q_bounds])


# ---- process bootstrap -------------------------------------------------------------------------------
def _rendezvous_key() -> str:
    """Names this job: the same string in every rank of one launch, different between launches.
    ``FFX_RDZV_KEY`` if set (bench.py's own launcher); else derived from what torch.distributed.run
    gives its workers (they share a parent process, a master port and a run id)."""
    key = os.environ.get("FFX_RDZV_KEY")
    if key:
        return key
    return "_".join(str(x) for x in (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                                     os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"), os.getppid()))


def _rendezvous_dir() -> str:
    """Where rank 0 publishes the communicator id: ``FFX_RDZV_DIR`` if set, else a directory of this user alone
    (mode 0700, owned by the effective uid -- checked, not assumed: the id names the shared-memory segment the other
    ranks will attach to, so nobody else may place or replace that file)."""
    given = os.environ.get("FFX_RDZV_DIR")
    if given:
        return given
    import stat

    problem = None
    for base in (os.environ.get("XDG_RUNTIME_DIR"), tempfile.gettempdir()):
        if not base:
            continue
        path = os.path.join(base, f"ffx-rdzv-{os.geteuid()}")
        try:
            os.mkdir(path, 0o700)
        except FileExistsError:
            pass
        except OSError as e:  # (a runtime directory that is announced but not writable: try the next place)
            problem = e
            continue
        st = os.lstat(path)
        if stat.S_ISDIR(st.st_mode) and st.st_uid == os.geteuid() and not (st.st_mode & 0o077):
            return path
        problem = PermissionError(f"{path} must be a directory owned by uid {os.geteuid()} with mode 0700")
    raise PermissionError(f"no private directory for the communicator id ({problem}); set FFX_RDZV_DIR to choose one")


def _read_rendezvous(path: str) -> bytes:
    """The id rank 0 published, or b"" if it is not there yet.  Only a fresh regular file owned by this user counts
    (no symlinks; a leftover of a crashed job is ignored)."""
    import stat

    try:
        fd = os.open(path, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
    except OSError:
        return b""
    try:
        st = os.fstat(fd)
        if not stat.S_ISREG(st.st_mode) or st.st_uid != os.geteuid() or time.time() - st.st_mtime >= 600:
            return b""
        return os.read(fd, 4096)
    finally:
        os.close(fd)


def connect(transport: str | int = "rccl", rank: int | None = None, world: int | None = None, device: int | None = None,
            key: str | None = None, timeout: float = 120.0):
    """Collective: build this job's communicator.  Rank 0 draws the id (``ffx_comm_unique_id``) and
    publishes it in a file named after the job's key under ``FFX_RDZV_DIR`` (default: a directory only this user
    can write, ``_rendezvous_dir``); the other ranks wait for that file (a regular file of this user, no symlinks).  ``rank`` / ``world`` / ``device`` default to RANK / WORLD_SIZE /
    LOCAL_RANK (what ``python -m torch.distributed.run`` and ``bench.py --gpus N`` export).

    ``transport``: "rccl" (one GPU per rank) or "host" (shared memory: ranks may share a GPU;
    ``device=-1`` for none at all); "rehearsal": this ONE process stands in for rank ``rank`` of
    ``world`` ranks that do not exist (timing of one rank's step on one GPU; not for results)."""
    from fast_forward_amd import native

    # one node by construction (one process per GPU of the same host): RCCL's bootstrap sockets can use
    # the loopback interface, which exists and resolves everywhere (container hostnames often do not)
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the driver supports dmabuf IPC only
    rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
    code = {"rccl": native.COMM_RCCL, "nccl": native.COMM_RCCL, "host": native.COMM_HOST, "rehearsal": native.COMM_REHEARSAL}.get(transport, transport)
    if code == native.COMM_REHEARSAL:  # no peers, nothing to rendezvous with
        return native.Comm(code, 0 if device is None else device, rank, world, native.Comm.unique_id(code))
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", str(rank)))
        if code == native.COMM_HOST:
            device %= max(1, native.device_count())
    if world == 1:
        return native.Comm(code, device, 0, 1, native.Comm.unique_id(code))
    path = os.path.join(_rendezvous_dir(), f"ffx_rdzv_{key or _rendezvous_key()}_{code}")
    if rank == 0:
        uid = native.Comm.unique_id(code)
        tmp = f"{path}.{os.getpid()}"
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)  # atomic: a reader sees all 128 bytes or no file
    else:
        t0, uid = time.monotonic(), b""
        while len(uid) != native.COMM_ID_BYTES:
            uid = _read_rendezvous(path)
            if len(uid) != native.COMM_ID_BYTES:
                if time.monotonic() - t0 > timeout:
                    raise TimeoutError(f"rank {rank}: rank 0 never published the communicator id at {path}")
                time.sleep(0.01)
    try:
        comm = native.Comm(code, device, rank, world, uid)
    finally:
        if rank == 0:
            try:
                os.unlink(path)  # ffx_comm_init is collective: everybody has read the id by now
            except OSError:
                pass
    return comm


# ---- one rank of the row-sharded index --------------------------------------------------------------------
class ShardedScorer:
    """One rank of a row-sharded index (``ffx_sharded``): local ``gather_dot`` + all-to-all of scores
    + owner-side merge, all native.

    Two HIP streams and two plans: everything that needs the CUs (scoring of batch i, then the merge + ordering
    of batch i-1) runs in order on the compute stream, only the exchange of batch i runs beside it on the
    communicator's stream, so back-to-back ``run`` calls hide the exchange and nothing competes with
    ``gather_dot`` for the CUs; ``sync`` / ``fetch`` finish the last batch's merge."""

    def __init__(self, shard, doc_bounds, comm):
        from fast_forward_amd import native

        self.shard, self.comm = shard, comm
        self.rank, self.world = comm.rank, comm.world
        self.doc_bounds = None if doc_bounds is None else np.asarray(doc_bounds, dtype=np.int64)
        self._s = native.Sharded(shard, comm, self.doc_bounds)

    def set_queries(self, queries: np.ndarray) -> None:
        self._s.set_queries(queries)  # replicated: Q*d*4 bytes (21 MB at Q=6980, d=768)

    def set_pairs(self, pair_off, pair_ids, sparse, mode: int, holder=None) -> None:
        try:
            self._s.set_pairs(pair_off, pair_ids, sparse, mode, holder=holder)
        except Exception as e:
            from fast_forward_amd import native

            if isinstance(e, native.NotFound):
                raise IndexError(f"pair {e.pair} names an ordinal outside the sharded index") from None
            raise

    def stage_owned(self, q_bounds, own_queries, own_pair_off, own_pair_ids, own_sparse, mode: int, own_holder=None) -> None:
        """A FRESH batch of which this rank hands over only the queries it owns (``ffx_sharded_stage_owned``): ~P/world
        candidates per rank go to its GPU, the ordinals travel to the holders in a first all-to-all.  May be called while
        the previous batch runs; ``run`` then runs the batch staged last."""
        try:
            self._s.stage_owned(q_bounds, own_queries, own_pair_off, own_pair_ids, own_sparse, mode, own_holder=own_holder)
        except Exception as e:
            from fast_forward_amd import native

            if isinstance(e, native.NotFound):
                raise IndexError(f"pair {e.pair} of this rank's block names an ordinal outside the sharded index" if e.pair >= 0
                                 else "a candidate owned by another rank names an ordinal outside the sharded index") from None
            raise

    def score_owned(self, q_bounds, own_queries, own_pair_off, own_pair_ids, own_sparse, mode: int, alpha: float = 0.0, exact_sum: bool = True,
                    sort: bool = False, n_slices: int = 0, own_holder=None, out=None):
        """Host arrays in, host arrays out (``ffx_sharded_score``): this rank's block, sliced and pipelined."""
        try:
            return self._s.score_owned(q_bounds, own_queries, own_pair_off, own_pair_ids, own_sparse, mode, alpha, exact_sum=exact_sum, sort=sort,
                                       n_slices=n_slices, own_holder=own_holder, out=out)
        except Exception as e:
            from fast_forward_amd import native

            if isinstance(e, native.NotFound):
                raise IndexError(f"pair {e.pair} of this rank's block names an ordinal outside the sharded index" if e.pair >= 0
                                 else "a candidate owned by another rank names an ordinal outside the sharded index") from None
            raise

    def stage_queries(self, all_queries) -> None:
        """ALL query vectors of the batch about to be staged, replicated on every rank (``ffx_sharded_stage_queries``)."""
        self._s.stage_queries(all_queries)

    def fetch_async(self, out, ff: bool = True, interp: bool = False, order: bool = False) -> None:
        self._s.fetch_async(out, ff=ff, interp=interp, order=order)

    def run(self, alpha: float, exact_sum: bool = True, sort: bool = False) -> None:
        self._s.run(alpha, exact_sum=exact_sum, sort=sort)

    def sync(self) -> None:
        self._s.sync()

    @property
    def q_lo(self) -> int:
        return self._s.q_lo

    @property
    def q_hi(self) -> int:
        return self._s.q_hi

    @property
    def own_first_pair(self) -> int:
        return self._s.own_first_pair

    @property
    def own_pairs(self) -> int:
        return self._s.own_pairs

    def fetch(self, order: bool = False, out=None):
        """(dense, interpolated|None, order|None) of the owned queries, first-stage order; ``out``: arrays to re-use."""
        return self._s.fetch(ff=True, interp=self._s.has_sparse, order=order, out=out)

    def stats(self) -> dict:
        return self._s.stats()

    def close(self) -> None:
        self._s.close()


# ---- the Index API over a sharded index ------------------------------------------------------------------
def _owner_of(doc_id: str | None, psg_id: str | None, world: int) -> int:
    """Rank holding a row: by document id (so a document's passages share a rank), else by
    passage id.  crc32, not hash(): identical in every process."""
    import zlib

    key = doc_id if doc_id is not None else psg_id
    return zlib.crc32(key.encode()) % world


class _AgreedIndexError(IndexError):
    """An ``IndexError`` every rank raises at the same collective step (one of them found the id missing and said so)."""


class _ShardedIndex:
    """What the sharded index classes share: the ``Index`` surface over SPMD ranks.  Subclasses say
    who holds each candidate (``_code_routing``) and own the shard."""

    comm = None
    local = None  # the object carrying mode / query encoder (an Index)
    exact_sum = True

    # ---- configuration mirrors the local index -----------------------------------------------------------
    @property
    def mode(self):
        return self.local.mode

    @mode.setter
    def mode(self, mode) -> None:
        self.local.mode = mode

    @property
    def dim(self):
        return self.local.dim

    @property
    def query_encoder(self):
        return self.local.query_encoder

    @query_encoder.setter
    def query_encoder(self, encoder) -> None:
        self.local.query_encoder = encoder

    @property
    def quantizer(self):
        return self.local.quantizer

    def encode_queries(self, queries):
        """``Index.encode_queries`` (base.py:58-72): every rank encodes the same queries."""
        return self.local.encode_queries(queries)

    def consolidate(self, placement_tries: int = -1) -> None:
        """``InMemoryIndex.consolidate`` on this rank's shard (not collective: every rank re-seats the rows it holds)."""
        fn = getattr(self.local, "consolidate", None)
        if fn is not None:
            fn(placement_tries)

    def _all_ids(self, mine: set) -> set:
        """The union of every rank's id strings (collective; the ids travel once per rank: one buffer of the UTF-8 bytes laid
        back to back plus their lengths -- an id may hold any character, a newline included)."""
        ordered = sorted(mine)
        parts = [x.encode() for x in ordered]
        blob = b"".join(parts)
        lengths = np.array([len(p) for p in parts], dtype=np.int64)
        sizes = self.comm.allgather_array(np.array([len(blob), len(parts)], dtype=np.int64))
        room_b, room_n = max(1, int(sizes[:, 0].max())), max(1, int(sizes[:, 1].max()))
        pad = np.zeros(room_n, dtype=np.int64)
        pad[:len(parts)] = lengths
        all_len = self.comm.allgather_array(pad)
        out: set = set()
        for r, part in enumerate(self.comm.allgather_bytes(blob.ljust(room_b, b"\0"))):
            at = 0
            for n in all_len[r, :int(sizes[r, 1])].tolist():
                out.add(part[at:at + n].decode())
                at += n
        return out

    @property
    def doc_ids(self) -> set:
        """All document ids, of every shard (``Index.doc_ids``; collective: every rank asks together)."""
        return self._all_ids(self.local.doc_ids)

    @property
    def psg_ids(self) -> set:
        """All passage ids, of every shard (``Index.psg_ids``; collective)."""
        return self._all_ids(self.local.psg_ids)

    def _holders_by_code(self, names: np.ndarray, packed=None) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
        """For the DISTINCT candidate ids of a ranking: (holder rank, ordinal on this rank where this rank is
        the holder, "no rank holds it" mask).  Collective (every rank calls it with the same ids, in the same
        order: a ranking's id numbering depends on its strings only).  Every rank knows only the ids of the rows it
        holds: it looks the names up in its own table, and one small control message -- the bit mask of what each
        rank holds -- tells everybody the holder of every id, or that nobody holds it."""
        ord_u = self.local._ids.lookup_optional(names, self.local.mode.value, packed)
        held = self.comm.allgather_array(np.packbits(ord_u >= 0))
        held = np.unpackbits(held, axis=1, count=len(names)).astype(bool) if len(names) else np.zeros((self.world, 0), dtype=bool)
        holder = held.argmax(axis=0).astype(np.int32) if len(names) else np.zeros(0, dtype=np.int32)
        return holder, np.maximum(ord_u, 0), ~held.any(axis=0)

    def _code_routing(self, ranking) -> dict:
        """What routing needs of a ranking's DISTINCT ids, looked up once per (ranking, index, mode) and remembered with
        the ranking (collective on a miss): ``holder_c[c]`` = rank holding the id with code c (-1: nobody) and
        ``ordinal_c[c]`` = its ordinal on THIS rank (0 where another rank holds it)."""
        codes, names = ranking._id_coding()
        key = (id(self), self.local.mode, getattr(self, "_n_global", None), getattr(self.local._ids, "version", None))
        cached = getattr(ranking, "_holder_cache", None)
        if cached is None or cached["key"] != key or cached["names"] is not names or cached["codes"] is not codes:
            holder_c, local_c, missing_c = self._holders_by_code(names, ranking._id_blob(names))
            cached = {"key": key, "names": names, "codes": codes, "holder_c": np.where(missing_c, -1, holder_c).astype(np.int32),
                      "ordinal_c": local_c.astype(np.int32), "missing": bool(missing_c.any())}
            ranking._holder_cache = cached
        return cached

    @staticmethod
    def _raise_first_missing(routing: dict) -> None:
        """``IndexError`` naming the first id in ROW order that no rank holds (the reference's util.py:38-39)."""
        codes, names = routing["codes"], routing["names"]
        bad = np.flatnonzero(routing["holder_c"][codes] < 0)
        raise IndexError(f"ID {names[codes[int(bad[0])]]} not found in the index.")

    def _streams_rows(self) -> bool:
        """True when this rank's rows are not resident and every call streams them from the file (``ShardedOnDiskIndex``)."""
        return False

    def _install_code_table(self, scorer: ShardedScorer, routing: dict) -> None:
        """This rank's ordinal of every code of the ranking (-1: held elsewhere) into the scorer's HBM, once per
        (ranking, index state): 4 bytes per DISTINCT id, where the candidate lists are 4 bytes per row."""
        if getattr(scorer, "_code_table_of", None) is routing:
            return
        table = np.where(routing["holder_c"] == self.comm.rank, routing["ordinal_c"], np.int32(-1)).astype(np.int32)
        scorer._s.set_code_table(table)
        scorer._code_table_of = routing  # (keeps the routing alive: `is` cannot be fooled by a recycled id)

    def _sharded(self) -> ShardedScorer:
        raise NotImplementedError

    def _scorer_queries(self, query_vectors: np.ndarray) -> np.ndarray:
        """The query vectors as the resident scorer wants them (an OPQ shard: rotated once, see InMemoryIndex)."""
        return query_vectors

    def _score_held(self, query_vectors, local_off, ordinals) -> np.ndarray:
        """Dense scores of candidates THIS rank holds (a CSR over all queries of the batch, ordinals of this rank), by the
        rank's own 1-GPU path: the resident shard, or -- a shard that stays in its file -- the streaming path."""
        return self.local._score_pairs(query_vectors, local_off, ordinals, self.local.mode)[0]

    def _rank(self, ranking, alpha: float | None, name, gather: bool):
        """Score, exchange, finish the owned queries; the ``Ranking`` of this rank's queries (or of all).  The same one
        route as ``Index.__call__``: integers in (``_Run``), a coded ``Ranking`` out."""
        from fast_forward_amd import native
        from fast_forward_amd.index.api import _Run

        if not ranking.has_queries:
            raise ValueError("Input ranking has no queries attached.")
        run = _Run(self.local, ranking)
        pair_off = run.pair_off
        mode = self.local.mode.value
        if not self._streams_rows():
            # Owner-side routing: this rank hands its GPU only the queries it OWNS (~P/world candidates).  Rows are placed
            # by a hash of the document id and every rank knows the ordinals of its own rows only, so the candidates travel
            # as the ranking's id CODES (every rank numbers a ranking's ids alike) beside their holder, and the holder
            # translates what arrives through a table in its HBM (ffx_sharded_set_code_table, set once per ranking).
            routing = self._code_routing(ranking)
            if routing["missing"]:
                self._raise_first_missing(routing)
            scorer = self._sharded()
            self._install_code_table(scorer, routing)
            qb = native.query_partition(run.n_q, self.comm.world)
            q_lo, q_hi = int(qb[self.comm.rank]), int(qb[self.comm.rank + 1])
            lo, hi = int(pair_off[q_lo]), int(pair_off[q_hi])
            rows = slice(lo, hi) if run.perm is None else run.perm[lo:hi]  # this block's rows of the ranking
            own_codes = np.ascontiguousarray(run.codes[rows], dtype=np.int32)
            own_sparse = np.ascontiguousarray(ranking._scores()[rows], dtype=np.float32) if alpha is not None else None
            # (an OPQ shard rotates the queries: ALL of them in one product, then the block -- a matrix product's rounding depends on
            # its shape, and the block of one query must get the bits the one-GPU index gets for that query)
            own_q = (np.ascontiguousarray(self._scorer_queries(run.query_vectors)[q_lo:q_hi]) if q_hi > q_lo
                     else np.zeros((0, scorer.shard.query_dim), dtype=np.float32))
            ff, interp, order = scorer.score_owned(qb, own_q, pair_off[q_lo:q_hi + 1] - lo, own_codes, own_sparse, mode, 0.0 if alpha is None else alpha,
                                                   exact_sum=self.exact_sum, sort=True, own_holder=routing["holder_c"][own_codes])
        else:
            # the shard is streamed from its file: the same owner-side routing on host buffers -- this rank takes in the block
            # of the queries it owns, the holders get the id codes of what they hold (and score them by the 1-GPU streaming
            # path), the scores come back; then the owner-side epilogue (ffx_aggregate, 1-row segments)
            routing = self._code_routing(ranking)
            if routing["missing"]:
                self._raise_first_missing(routing)
            qb = native.query_partition(run.n_q, self.comm.world)
            q_lo, q_hi = int(qb[self.comm.rank]), int(qb[self.comm.rank + 1])
            lo, hi = int(pair_off[q_lo]), int(pair_off[q_hi])
            rows = slice(lo, hi) if run.perm is None else run.perm[lo:hi]  # this block's rows of the ranking
            own_codes = np.ascontiguousarray(run.codes[rows], dtype=np.int32)
            dense = self._owner_routed_dense(run.query_vectors, qb, pair_off[q_lo:q_hi + 1] - lo, own_codes, routing)
            if hi > lo:
                own_sparse = np.ascontiguousarray(ranking._scores()[rows], dtype=np.float32) if alpha is not None else None
                ff, interp, order = native.aggregate(dense, np.arange(hi - lo + 1, dtype=np.int64), 1, own_sparse,
                                                     0.0 if alpha is None else alpha, pair_off[q_lo:q_hi + 1] - lo, sort=True, device=self.comm.device)
            else:
                ff, interp, order = dense, (dense if alpha is not None else None), np.zeros(0, dtype=np.int32)
        score = interp if alpha is not None else ff
        # `order`: positions inside the owned block, every query ordered -> positions in the grouped layout
        kept, kept_scores = order.astype(np.int32, copy=False) + np.int32(lo), native.take32(score, order).astype(np.float32, copy=False)
        if not gather:
            return run.result(name, kept, kept_scores, pair_off[q_lo:q_hi + 1] - lo, q_range=(q_lo, q_hi))[0]
        # every rank's block, in rank order = query order (queries are owned in contiguous blocks)
        sizes = self.comm.allgather_array(np.array([hi - lo], dtype=np.int64))[:, 0]
        room = int(sizes.max())
        pad_k, pad_s = np.zeros(room, dtype=np.int32), np.zeros(room, dtype=np.float32)
        pad_k[:hi - lo], pad_s[:hi - lo] = kept, kept_scores
        all_k, all_s = self.comm.allgather_array(pad_k), self.comm.allgather_array(pad_s)
        full_k = np.concatenate([all_k[r, :int(sizes[r])] for r in range(self.comm.world)])
        full_s = np.concatenate([all_s[r, :int(sizes[r])] for r in range(self.comm.world)])
        return run.result(name, full_k, full_s, pair_off)[0]

    def _owner_routed_dense(self, query_vectors, q_bounds, own_off, own_codes, routing, error: "str | None" = None) -> np.ndarray:
        """Dense scores of the candidates of the queries THIS rank owns (collective; the host-buffer form of
        ``ffx_sharded_stage_owned`` + run, for shards that are streamed and for the steps of early stopping).

        ``own_off`` / ``own_codes``: CSR over this rank's block ``q_bounds[rank] .. q_bounds[rank + 1]`` of the batch's queries,
        candidates as the ranking's id codes (every rank numbers a ranking's ids alike; ``routing`` names each code's holder
        and its ordinal there).  A rank takes in its own block and what the owners send it -- the codes of the candidates
        it holds, with their per-query counts -- never the batch: per-rank intake is proportional to 1 / world
        (``last_intake`` says what it was).  Every holder scores what it received with its own 1-GPU path (resident or
        streamed) and returns the scores to the owners, which put them back in candidate order.

        ``error``: a message this rank wants EVERY rank to raise as ``IndexError`` at this step instead (an unknown id
        reached by one owner's early-stopping loop must not leave the others waiting in the exchange)."""
        comm, world, me = self.comm, self.comm.world, self.comm.rank
        q_bounds = np.asarray(q_bounds, dtype=np.int64)
        own_off = np.asarray(own_off, dtype=np.int64)
        n_own_q, n_own = own_off.shape[0] - 1, int(own_off[-1])
        assert n_own_q == int(q_bounds[me + 1] - q_bounds[me])
        own_codes = np.ascontiguousarray(own_codes, dtype=np.int32)
        holder = routing["holder_c"][own_codes] if n_own else np.zeros(0, dtype=np.int32)
        if error is None and n_own and int(holder.min()) < 0:
            error = f"ID {routing['names'][own_codes[int(np.flatnonzero(holder < 0)[0])]]} not found in the index."
        message = (error or "").encode()[:400]
        counts = np.bincount(holder, minlength=world).astype(np.int64) if error is None else np.zeros(world, dtype=np.int64)
        header = comm.allgather_array(np.concatenate([[len(message)], counts]).astype(np.int64))  # (world, 1 + world)
        if header[:, 0].any():
            said = comm.allgather_bytes(message.ljust(400, b"\0"))
            raise _AgreedIndexError(next(m.rstrip(b"\0").decode(errors="replace") for m, n in zip(said, header[:, 0]) if n))
        # to every holder: how many of each of my queries' candidates it holds, then their codes (in my candidate order)
        by_holder = np.argsort(holder, kind="stable")
        q_of = np.repeat(np.arange(n_own_q, dtype=np.int64), np.diff(own_off))
        per_query = (np.bincount(holder.astype(np.int64) * n_own_q + q_of, minlength=world * n_own_q).astype(np.int32).reshape(world, n_own_q)
                     if n_own_q else np.zeros((world, 0), dtype=np.int32))
        starts = np.concatenate([[0], np.cumsum(counts)])
        grouped = own_codes[by_holder]
        send = np.concatenate([part for t in range(world) for part in (per_query[t], grouped[starts[t]:starts[t + 1]])]).astype(np.int32, copy=False)
        q_per_rank = np.diff(q_bounds)
        held_from = header[:, 1 + me]  # candidates rank r sends me
        got = comm.exchange_host(send.view(np.float32), n_own_q + counts, q_per_rank + held_from).view(np.int32)
        # what arrived, owner after owner = query after query: the CSR of the candidates this rank holds, as ordinals
        at, local_counts, held_codes = 0, [], []
        for r in range(world):
            local_counts.append(got[at:at + int(q_per_rank[r])])
            at += int(q_per_rank[r])
            held_codes.append(got[at:at + int(held_from[r])])
            at += int(held_from[r])
        local_off = np.zeros(int(q_bounds[-1]) + 1, dtype=np.int64)
        np.cumsum(np.concatenate(local_counts), out=local_off[1:])
        held_codes = np.concatenate(held_codes)
        self.last_intake = {"own_pairs": n_own, "held_pairs": int(held_codes.shape[0]), "batch_pairs": int(header[:, 1:].sum()),
                            "bytes": int(own_off.nbytes + own_codes.nbytes + got.nbytes)}
        if held_codes.shape[0]:
            ordinals = np.ascontiguousarray(routing["ordinal_c"][held_codes], dtype=np.int32)
            scores = np.ascontiguousarray(self._score_held(query_vectors, local_off, ordinals), dtype=np.float32)
        else:
            scores = np.zeros(0, dtype=np.float32)
        back = comm.exchange_host(scores, held_from, counts)  # owner r's candidates stand together (its queries do), in its order
        dense = np.empty(n_own, dtype=np.float32)
        dense[by_holder] = back
        return dense

    def _early_stopping(self, ranking, cutoff: int, alpha: float, depths, gather: bool):
        """``Index._early_stopping`` (base.py:316-387) over the shards.  The reference's decisions are per query, so every
        rank runs the loop over depth intervals (``Index._early_stopping_kept``) for the queries it OWNS, and scoring an
        interval is one owner-routed step (``_owner_routed_dense``): a rank takes in its block and what it holds of the
        others', never the batch.  Every rank joins the same number of steps -- one per depth at or beyond the cutoff; a
        rank whose queries are done joins the rest with empty lists -- and an unknown id that one owner's loop reaches is
        raised on every rank at that step.  Returns the ranking of the candidates that were scored: of this rank's owned
        queries, or (``gather``) of all queries on every rank."""
        from fast_forward_amd import native
        from fast_forward_amd.index.api import Index, _Run

        run = _Run(self.local, ranking)
        off = run.pair_off
        routing = self._code_routing(ranking)
        world, me = self.comm.world, self.comm.rank
        qb = native.query_partition(run.n_q, world)
        q_lo, q_hi = int(qb[me]), int(qb[me + 1])
        lo, hi = int(off[q_lo]), int(off[q_hi])
        rows_of_block = slice(lo, hi) if run.perm is None else run.perm[lo:hi]
        own_codes = np.ascontiguousarray(run.codes[rows_of_block], dtype=np.int32)
        own_sparse = np.ascontiguousarray(ranking._scores()[rows_of_block])
        own_off = off[q_lo:q_hi + 1] - lo
        steps = sum(1 for b in depths if b >= cutoff)  # the most intervals any rank's loop can score
        taken = [0]
        outer = self

        class Scored:  # what Index._early_stopping_kept needs of an index
            mode = self.local.mode

            def _es_bound_f32(self):
                return outer.local._es_bound_f32()

            def _score_pairs(self, query_vectors, part_off, positions, mode):
                taken[0] += 1
                return outer._owner_routed_dense(run.query_vectors, qb, part_off, own_codes[positions], routing), None, None

        known = routing["holder_c"][own_codes] >= 0 if hi > lo else np.zeros(0, dtype=bool)
        positions = np.where(known, np.arange(hi - lo, dtype=np.int64), -1)  # -1: no rank holds the id (an error if reached)
        failure, rows, scores = None, np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.float32)
        try:
            rows, scores = Index._early_stopping_kept(Scored(), lambda p: run.names[own_codes[p]], own_sparse, own_off, run.query_vectors[q_lo:q_hi], cutoff,
                                                      alpha, depths, positions)
        except _AgreedIndexError:
            raise  # (every rank is raising this at the same step)
        except IndexError as e:
            if taken[0] >= steps:
                raise
            failure = str(e)  # reached HERE, between two steps: the next step tells every rank, and every rank raises there
        empty_off = np.zeros(q_hi - q_lo + 1, dtype=np.int64)
        while taken[0] < steps:  # this rank's queries are done (or it has something to say): the others' steps still need it
            taken[0] += 1
            self._owner_routed_dense(run.query_vectors, qb, empty_off, np.zeros(0, dtype=np.int32), routing, error=failure)
        rows, scores = rows.astype(np.int32), np.ascontiguousarray(scores, dtype=np.float32)
        kept_off = np.searchsorted(rows, own_off).astype(np.int64)
        order = native.host_order(scores, kept_off)
        kept, kept_scores = native.take32(rows, order) + np.int32(lo), native.take32(scores, order)
        if not gather:
            return run.result("fast-forward", kept, kept_scores, kept_off, q_range=(q_lo, q_hi))[0]
        # every rank's block, in rank order = query order (queries are owned in contiguous blocks)
        sizes = self.comm.allgather_array(np.array([kept.shape[0]], dtype=np.int64))[:, 0]
        most_q = int(np.diff(qb).max())
        counts_all = self.comm.allgather_array(np.pad(np.diff(kept_off), (0, most_q - (q_hi - q_lo))).astype(np.int64))
        room = max(1, int(sizes.max()))
        pad_k, pad_s = np.zeros(room, dtype=np.int32), np.zeros(room, dtype=np.float32)
        pad_k[:kept.shape[0]], pad_s[:kept.shape[0]] = kept, kept_scores
        all_k, all_s = self.comm.allgather_array(pad_k), self.comm.allgather_array(pad_s)
        full_k = np.concatenate([all_k[r, :int(sizes[r])] for r in range(world)])
        full_s = np.concatenate([all_s[r, :int(sizes[r])] for r in range(world)])
        full_off = np.zeros(run.n_q + 1, dtype=np.int64)
        np.cumsum(np.concatenate([counts_all[r, :int(qb[r + 1] - qb[r])] for r in range(world)]), out=full_off[1:])
        return run.result("fast-forward", full_k, full_s, full_off)[0]

    def __call__(self, ranking, early_stopping: int | None = None, early_stopping_alpha: float | None = None,
                 early_stopping_depths=None, batch_size: int | None = None, gather: bool = False):
        """Dense scores (``Index.__call__``, base.py:389-396: same keywords) of the queries this rank owns, or of all
        with ``gather``.  ``batch_size`` is accepted and has no effect (nothing is materialised per batch here)."""
        if not ranking.has_queries:
            raise ValueError("Input ranking has no queries attached.")
        if early_stopping is not None:
            if early_stopping_alpha is None or early_stopping_depths is None:
                raise ValueError("Early stopping requires alpha and depths.")
            return self._early_stopping(ranking, early_stopping, early_stopping_alpha, early_stopping_depths, gather)
        return self._rank(ranking, None, "fast-forward", gather)

    def rerank(self, ranking, alpha: float, gather: bool = False):
        """``ranking.interpolate(index(ranking), alpha)`` with the epilogue on the owners' devices."""
        return self._rank(ranking, alpha, ranking.name, gather)


class ShardedInMemoryIndex(_ShardedIndex):
    """``InMemoryIndex`` spread over the ranks of a communicator (SPMD: every rank makes the same
    calls with the same arguments).

    * ``add(vectors, doc_ids, psg_ids)``: each rank keeps the rows whose document (or, for rows
      without a document id, passage) hashes to it, in its own HBM.
    * ``__call__(ranking)`` / ``rerank(ranking, alpha)``: every rank resolves the candidate ids it
      holds (an id held by no rank is the reference's ``IndexError``, raised on all ranks), scores
      them locally, and the dense scores of a query travel to the rank that owns the query (one
      all-to-all, ``ffx_comm_exchange``).  Each rank returns the ``Ranking`` of the queries it owns;
      ``gather=True`` assembles the complete ``Ranking`` on every rank.

    Scores equal a single ``InMemoryIndex`` holding all rows, bit for bit (same kernels, same
    per-document row order)."""

    def __init__(self, query_encoder=None, mode=None, encoder_batch_size: int = 32, comm=None, exact_sum: bool = True, **index_kwargs):
        from fast_forward_amd.index import InMemoryIndex, Mode

        self.comm = comm if comm is not None else connect()
        self.rank, self.world, self.device = self.comm.rank, self.comm.world, self.comm.device
        self.exact_sum = exact_sum
        self.local = InMemoryIndex(query_encoder, mode=mode or Mode.MAXP, encoder_batch_size=encoder_batch_size, device=self.device,
                                   exact_sum=exact_sum, **index_kwargs)
        self._n_global = 0
        self._scorer: ShardedScorer | None = None
        self._scorer_store = None

    def __len__(self) -> int:
        return self._n_global

    def add(self, vectors: np.ndarray, doc_ids=None, psg_ids=None) -> None:
        n = vectors.shape[0]
        doc_ids = [None] * n if doc_ids is None else doc_ids
        psg_ids = [None] * n if psg_ids is None else psg_ids
        if not len(doc_ids) == len(psg_ids) == n:
            raise ValueError("Number of IDs does not match number of vectors.")
        if any(d is None and p is None for d, p in zip(doc_ids, psg_ids)):
            raise ValueError("Vector has neither document nor passage ID.")
        mine = [i for i in range(n) if _owner_of(doc_ids[i], psg_ids[i], self.world) == self.rank]
        if self._scorer is not None:  # its plans belong to the store, which add() may re-allocate (on every rank alike)
            self._scorer.close()
            self._scorer = self._scorer_store = None
            self._drop_placeholder()
        if mine:
            self.local.add(vectors[mine], doc_ids=[doc_ids[i] for i in mine], psg_ids=[psg_ids[i] for i in mine])
        self._n_global += n

    # ---- scoring ---------------------------------------------------------------------------------------------
    def _sharded(self) -> ShardedScorer:
        """The native scorer over the local store (rebuilt after ``add``, which every rank calls alike, so every rank
        rebuilds in the same call).  A rank whose hash bucket is empty -- crc32 placement of a small collection over
        8 ranks easily leaves one -- takes part with a one-row placeholder store and empty candidate lists: what a
        rank does about "no rows here" is decided TOGETHER (one small all-gather), never by one rank raising between
        two collectives while the others wait for it in the exchange."""
        from fast_forward_amd import native

        if self._scorer is not None:
            return self._scorer
        store = self.local._store if len(self.local) else None
        has_book = self.local._quantizer is not None and self.local._quantizer.device_codebook() is not None
        mine = np.array([len(self.local), 0 if store is None else store.dim, -1 if store is None else native._DTYPE_CODE[np.dtype(store.dtype)],
                         1 if has_book else 0], dtype=np.int64)
        everyone = self.comm.allgather_array(mine)
        if int(everyone[:, 0].max()) == 0:
            raise RuntimeError("the sharded index is empty on every rank")  # every rank raises: nobody is left waiting
        # a rank without rows takes part with a placeholder store of the others' kind; for product-quantized rows it needs the
        # codebook for that.  Whether every such rank has one is known to all (the gathered vector): all raise, or none
        quantized = bool((everyone[:, 2] == native._DTYPE_CODE[np.dtype(np.uint8)]).any())
        if quantized and bool(((everyone[:, 0] == 0) & (everyone[:, 3] == 0)).any()):
            raise RuntimeError("a rank without rows needs the index's quantizer to take part in a quantized sharded index")
        if store is None:
            donor = everyone[int(np.argmax(everyone[:, 0] > 0))]
            dtype = next(dt for dt, code in native._DTYPE_CODE.items() if code == int(donor[2]))
            store = native.DeviceIndex(1, int(donor[1]), self.device, dtype=dtype)
            if dtype == np.uint8:  # product-quantized rows: the kernels are picked from the codebook's shape
                book = self.local._quantizer.device_codebook() if self.local._quantizer is not None else None
                if book is None:
                    raise RuntimeError("a rank without rows needs the index's quantizer to take part in a quantized sharded index")
                store.set_codebook(book[0])
            store.set_doc_rows(np.zeros(1, dtype=np.int64), np.zeros(0, dtype=np.int32))
            store.set_psg_rows(np.zeros(0, dtype=np.int32))
            self._placeholder = store
        else:
            self.local._sync_ids()
        self._scorer, self._scorer_store = ShardedScorer(store, None, self.comm), store
        return self._scorer

    def _drop_placeholder(self) -> None:
        if getattr(self, "_placeholder", None) is not None:
            self._placeholder.close()
            self._placeholder = None

    def close(self) -> None:
        if self._scorer is not None:
            self._scorer.close()
            self._scorer = None
        self._drop_placeholder()
        self.local.close()


class ShardedOnDiskIndex(_ShardedIndex):
    """An index FILE (this build's container, or an HDF5 file written by the reference) spread over
    the ranks of a communicator: BASELINE config 5 (100 M x 768 fp32 = 307 GB does not fit one GPU;
    over 8 GPUs a shard is 38.4 GB and is simply resident).

    Every rank opens the same file as a SHARD VIEW (``OnDiskIndex.load(..., shard=(rank, world))``): a row belongs to
    the rank its document id (its passage id, if it has no document id) hashes to -- crc32 % world, the placement
    ``ShardedInMemoryIndex`` uses, so a document's passages share a rank -- and a rank reads the file's id columns
    block by block, keeps and interns ONLY the ids of its own rows (host memory ~ N / world per rank; the reference,
    one process, reads all N: disk.py:355-418), and moves ITS rows from the chunk maps to its HBM through
    ``ffx_index_upload_rows`` (native threads, page-locked staging) -- the sharded form of ``OnDiskIndex.to_memory``
    (disk.py:177-205).  Who holds a candidate is settled per call as in ``ShardedInMemoryIndex``: every rank looks the
    ranking's distinct ids up in its own table and one small collective (a bit mask per rank) tells everybody the
    holder, or that nobody holds the id (``IndexError`` on every rank).  Scoring is then exactly the resident sharded
    path (local gather_dot, one all-to-all of scores, owner-side merge).

    ``resident=False`` (or a shard larger than the free HBM): the shard stays in the file and each
    call streams the rows its local candidates need through two HBM windows, as the 1-GPU
    ``OnDiskIndex`` does (reference disk.py:309-336); local scores are then exchanged from host
    buffers.  Either way the scores equal the single-process result bit for bit."""

    def __init__(self, index_file, comm=None, query_encoder=None, mode=None, encoder_batch_size: int = 32, exact_sum: bool = True,
                 resident: bool | None = None, window_bytes: int = 1 << 30):
        from fast_forward_amd import native
        from fast_forward_amd.index import Mode
        from fast_forward_amd.index.ondisk import _READ_THREADS, OnDiskIndex

        self.comm = comm if comm is not None else connect()
        self.rank, self.world, self.device = self.comm.rank, self.comm.world, self.comm.device
        self.exact_sum = exact_sum
        self.local = OnDiskIndex.load(index_file, query_encoder=query_encoder, mode=mode or Mode.MAXP, encoder_batch_size=encoder_batch_size,
                                      device=self.device, exact_sum=exact_sum, window_bytes=window_bytes, shard=(self.rank, self.world))
        disk, ids = self.local, self.local._ids
        n_local = len(disk)
        self._n_file = int(disk._meta["num_vectors"])
        self.rows_per_rank = self.comm.allgather_array(np.array([n_local], dtype=np.int64))[:, 0]
        stored = np.dtype(disk._meta["dtype"]) if disk._meta["dtype"] else np.dtype(np.float32)
        book = disk._quantizer.device_codebook() if disk._quantizer is not None else None
        direct = (stored in (np.dtype(np.float32), np.dtype(np.float16), np.dtype(np.uint8)) and (disk._quantizer is None) == (stored != np.dtype(np.uint8))
                  and (disk._quantizer is None or book is not None))
        shard_bytes = n_local * int(disk._meta["dim"] or 0) * (stored.itemsize if direct else 4)
        if resident is None:
            resident = shard_bytes <= 0.8 * native.device_info(self.device)["free_bytes"]
        # every rank must take the same branch (the exchange differs): resident only if it is for all
        self.resident = bool(self.comm.allreduce([1.0 if resident else 0.0], native.OP_MIN)[0])
        self._store = self._scorer = None
        self._rotation = None
        # (every rank holds the same rows_per_rank, so what follows is the same decision everywhere: nobody raises alone
        # between two collectives)
        if int(self.rows_per_rank.sum()) == 0 and self.resident:
            raise RuntimeError("the index file holds no rows")
        if not self.resident:
            return
        quantized = disk._quantizer is not None and book is not None and direct
        if n_local:
            chunk_maps = [m[0] for m in disk._maps]
            if direct and all(isinstance(c, np.ndarray) and c.flags["C_CONTIGUOUS"] for c in chunk_maps):
                self._store = native.DeviceIndex(n_local, int(disk._meta["dim"]), self.device, dtype=stored)
                if book is not None:
                    self._store.set_codebook(book[0])
                self._store.upload_rows(chunk_maps, disk._row_map, 0, _READ_THREADS)
            else:  # float64 files of the reference, or a quantizer the kernels cannot read: decoded fp32 rows
                data = disk._read_rows(np.arange(n_local))
                if disk._quantizer is not None:
                    data = disk._quantizer.decode(data)
                self._store = native.DeviceIndex(n_local, data.shape[1], self.device, dtype=np.float32)
                self._store.upload(np.ascontiguousarray(data, dtype=np.float32), 0)
            self._rotation = book[1] if (book is not None and self._store.dtype == np.uint8) else None
            doc_off, doc_rows = ids.doc_csr()  # rows of a document: local row numbers, insertion order
            self._store.set_doc_rows(doc_off, doc_rows)
            self._store.set_psg_rows(ids.psg_rows())
        else:
            # this rank's hash bucket is empty (fewer documents than ranks): it takes part with a one-row placeholder and
            # empty candidate lists
            self._store = native.DeviceIndex(1, int(disk._meta["dim"]) if direct else int(disk.dim), self.device, dtype=stored if direct else np.float32)
            if quantized:
                self._store.set_codebook(book[0])
            self._rotation = book[1] if quantized else None
            self._store.set_doc_rows(np.zeros(1, dtype=np.int64), np.zeros(0, dtype=np.int32))
            self._store.set_psg_rows(np.zeros(0, dtype=np.int32))
        self._scorer = ShardedScorer(self._store, None, self.comm)

    def __len__(self) -> int:
        return self._n_file

    def _sharded(self) -> ShardedScorer:
        return self._scorer

    def _scorer_queries(self, query_vectors):
        # OPQ decodes to c @ R.T, and <q, c R^T> = <q R, c>: the queries are rotated once (as InMemoryIndex does)
        rotation = getattr(self, "_rotation", None)
        return query_vectors if rotation is None else np.ascontiguousarray(np.asarray(query_vectors, dtype=np.float32) @ rotation, dtype=np.float32)

    def _streams_rows(self) -> bool:
        return not self.resident

    def _score_held(self, query_vectors, local_off, ordinals) -> np.ndarray:
        if not self.resident:  # the shard view numbers its documents / passages itself: the 1-GPU streaming path takes it from there
            return self.local._score_pairs(query_vectors, local_off, ordinals, self.local.mode)[0]
        from fast_forward_amd import native

        return native.score(self._store, self._scorer_queries(query_vectors), local_off, ordinals, self.local.mode.value, None, 0.0,
                            exact_sum=self.exact_sum)[0]

    def close(self) -> None:
        if self._scorer is not None:
            self._scorer.close()
            self._scorer = None
        if self._store is not None:
            self._store.close()
            self._store = None
        self.local._release_windows()
