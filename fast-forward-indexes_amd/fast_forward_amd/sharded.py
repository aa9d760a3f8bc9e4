"""Row-sharded index over the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference is single-process and has no counterpart; this is the multi-GPU form of the same
path.  The embedding matrix is sharded by *document* (a document's
passages never straddle ranks, so MaxP/FirstP/AveP stay rank-local), query vectors are
replicated, and every rank scores the candidates whose documents it holds.  The one exchange
step is an all-to-all of dense scores: query ``q`` is *owned* by one rank, which receives q's
scores from every rank, puts them back in first-stage order, interpolates and orders them
(``ffx_merge_*``).  Per step each rank sends/receives ~P/world * 4 bytes, spread over all
peers at once -- with 7 point-to-point xGMI links per GPU a full-mesh all-to-all uses every
link, where a ring all-gather of all scores would be bound by one.

Routing (which rank scores which candidate, where each score lands at the owner) is computed on
the host once per run by every rank from the same CSR; it is pure NumPy and is covered by the
world_size-2 gloo tests, which drive exactly these functions with CPU tensors.
"""

from __future__ import annotations

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


def exchange(send, send_counts, recv_counts, group=None):
    """All-to-all of dense scores (torch tensors, CPU/gloo or GPU/RCCL): slice j of ``send`` goes
    to rank j; returns the concatenation of what every rank sent here, in rank order."""
    import torch
    import torch.distributed as dist

    recv = torch.empty(int(np.sum(recv_counts)), dtype=send.dtype, device=send.device)
    dist.all_to_all_single(recv, send, [int(c) for c in recv_counts], [int(c) for c in send_counts], group=group)
    return recv


class ShardedScorer:
    """One rank of a row-sharded index: local ``gather_dot`` + RCCL all-to-all + owner merge.

    Two HIP streams and two plans: batch i is scored on the compute stream while the scores of
    batch i-1 are still being exchanged and merged on the communication stream, so back-to-back
    ``run`` calls cost max(scoring, exchange + merge) instead of their sum.  Each plan owns its
    send buffer; a plan is reused only after the exchange that read it has finished."""

    def __init__(self, shard, doc_bounds, rank: int, world: int, device: int, group=None):
        import torch  # before the first libffx call: see native.lib() on the one-runtime-per-process rule

        from fast_forward_amd import native

        native.lib()
        if not native.TORCH_LOADED_FIRST:
            raise RuntimeError("libffx was loaded before torch in this process; import torch first "
                               "(two HIP runtimes cannot share a process)")
        self._torch = torch
        self._native = native
        self.shard, self.doc_bounds, self.rank, self.world, self.device, self.group = shard, doc_bounds, rank, world, device, group
        self.plans = [native.Plan(shard), native.Plan(shard)]
        self.plan = self.plans[0]  # the plan of the most recent run (stats)
        self.merge = native.Merge(device)
        self.route: Route | None = None
        # explicit (non-null) streams: the C ABI reads a NULL stream as "use your own", so torch's
        # default stream (handle 0) is never used here
        self._compute, self._comm = torch.cuda.Stream(device), torch.cuda.Stream(device)
        assert self._compute.cuda_stream != 0 and self._comm.cuda_stream != 0
        for plan in self.plans:
            plan.set_stream(self._compute.cuda_stream)
        self._scored = [torch.cuda.Event(), torch.cuda.Event()]  # plan i has written its scores
        self._sent = [torch.cuda.Event(), torch.cuda.Event()]    # the exchange has read plan i's scores
        self._runs = 0

    def set_queries(self, queries: np.ndarray) -> None:
        self.sync()
        for plan in self.plans:
            plan.set_queries(queries)  # replicated: Q*d*4 bytes (21 MB at Q=6980, d=768)

    def set_pairs(self, pair_off, pair_ids, sparse, mode: int) -> None:
        torch, native = self._torch, self._native
        self.sync()
        r = route(pair_off, pair_ids, self.doc_bounds, self.rank, self.world, sparse)
        self._send = []
        for plan in self.plans:
            plan.set_pairs(r.local_pair_off, r.local_pair_ids, None, mode)
            self._send.append(torch.as_tensor(plan.device_array(native.BUF_FF), device=f"cuda:{self.device}"))
        self.merge.set(r.own_pair_off, r.recv_slot, r.own_sparse)
        with torch.cuda.stream(self._comm):
            self._recv = torch.empty(int(r.recv_counts.sum()), dtype=torch.float32, device=f"cuda:{self.device}")
        self._in_split = [int(c) for c in r.send_counts]
        self._out_split = [int(c) for c in r.recv_counts]
        self.route = r

    def run(self, alpha: float, exact_sum: bool = True, sort: bool = False) -> None:
        import torch.distributed as dist

        torch = self._torch
        i = self._runs & 1
        self._runs += 1
        plan, send = self.plans[i], self._send[i]
        self.plan = plan
        with torch.cuda.stream(self._compute):
            self._compute.wait_event(self._sent[i])  # the exchange of two runs ago read this plan's scores
            plan.run(0.0, exact_sum=exact_sum)
            self._scored[i].record(self._compute)
        with torch.cuda.stream(self._comm):
            self._comm.wait_event(self._scored[i])
            if dist.get_backend(self.group) == "gloo":
                # rehearsal path (several ranks sharing one GPU, where RCCL refuses to run): the
                # same exchange staged through host memory
                self._comm.synchronize()
                recv_host = torch.empty(self._recv.shape[0], dtype=torch.float32)
                dist.all_to_all_single(recv_host, send.cpu(), self._out_split, self._in_split, group=self.group)
                self._recv.copy_(recv_host)
            else:
                dist.all_to_all_single(self._recv, send, self._out_split, self._in_split, group=self.group)
            self._sent[i].record(self._comm)
            self.merge.run(self._recv.data_ptr(), alpha, sort=sort, stream=self._comm.cuda_stream)

    def sync(self) -> None:
        self._torch.cuda.synchronize(self.device)

    def fetch(self, order: bool = False):
        """(dense, interpolated|None, order|None) of the owned queries, first-stage order."""
        return self.merge.fetch(ff=True, interp=self.merge.has_sparse, order=order)

    def close(self) -> None:
        self.sync()
        for plan in self.plans:
            plan.close()
        self.merge.close()


# ---- the Index API over a sharded index ------------------------------------------------------------------
def _owner_of(doc_id: str | None, psg_id: str | None, world: int) -> int:
    """Rank holding a row: by document id (so a document's passages share a rank), else by
    passage id.  crc32, not hash(): identical in every process."""
    import zlib

    key = doc_id if doc_id is not None else psg_id
    return zlib.crc32(key.encode()) % world


class ShardedInMemoryIndex:
    """``InMemoryIndex`` spread over the ranks of a ``torch.distributed`` group (SPMD: every rank
    makes the same calls with the same arguments).

    * ``add(vectors, doc_ids, psg_ids)``: each rank keeps the rows whose document (or, for rows
      without a document id, passage) hashes to it, in its own HBM.
    * ``__call__(ranking)`` / ``rerank(ranking, alpha)``: every rank resolves the candidate ids it
      holds (an id held by no rank is the reference's ``IndexError``, raised on all ranks), scores
      them locally, and the dense scores of a query travel to the rank that owns the query
      (``exchange``: one all-to-all).  Each rank returns the ``Ranking`` of the queries it owns;
      ``gather=True`` assembles the complete ``Ranking`` on every rank.

    Scores equal a single ``InMemoryIndex`` holding all rows, bit for bit (same kernels, same
    per-document row order)."""

    def __init__(self, query_encoder=None, mode=None, encoder_batch_size: int = 32, device: int | None = None,
                 exact_sum: bool = True, group=None, **index_kwargs):
        import torch  # before libffx is loaded (native.lib)
        import torch.distributed as dist

        from fast_forward_amd.index import InMemoryIndex, Mode

        self._torch, self._dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if device is None:
            device = self.rank % max(1, torch.cuda.device_count())
        self.device = device
        if dist.get_backend(group) != "gloo":
            torch.cuda.set_device(device)  # object collectives of the RCCL backend use the current device
        self.local = InMemoryIndex(query_encoder, mode=mode or Mode.MAXP, encoder_batch_size=encoder_batch_size, device=device,
                                   exact_sum=exact_sum, **index_kwargs)
        self._n_global = 0

    # ---- configuration mirrors the local index -----------------------------------------------------------
    @property
    def mode(self):
        return self.local.mode

    @mode.setter
    def mode(self, mode) -> None:
        self.local.mode = mode

    def __len__(self) -> int:
        return self._n_global

    @property
    def dim(self):
        return self.local.dim

    def add(self, vectors: np.ndarray, doc_ids=None, psg_ids=None) -> None:
        n = vectors.shape[0]
        doc_ids = [None] * n if doc_ids is None else doc_ids
        psg_ids = [None] * n if psg_ids is None else psg_ids
        if not len(doc_ids) == len(psg_ids) == n:
            raise ValueError("Number of IDs does not match number of vectors.")
        if any(d is None and p is None for d, p in zip(doc_ids, psg_ids)):
            raise ValueError("Vector has neither document nor passage ID.")
        mine = [i for i in range(n) if _owner_of(doc_ids[i], psg_ids[i], self.world) == self.rank]
        if mine:
            self.local.add(vectors[mine], doc_ids=[doc_ids[i] for i in mine], psg_ids=[psg_ids[i] for i in mine])
        self._n_global += n

    # ---- scoring ---------------------------------------------------------------------------------------------
    def _score(self, ranking, alpha: float | None, sort: bool):
        import pandas as pd

        from fast_forward_amd.index.api import _grouped_by_query

        dist, torch = self._dist, self._torch
        if not ranking.has_queries:
            raise ValueError("Input ranking has no queries attached.")
        df = ranking._df
        q_no, q_ids, texts = self.local._numbered_queries(ranking)
        n_q = len(texts)
        query_vectors = np.ascontiguousarray(self.local.encode_queries(texts), dtype=np.float32)
        perm, pair_off = _grouped_by_query(q_no, n_q)
        ids = df["id"].to_numpy()
        sparse = df["score"].to_numpy(dtype=np.float32)
        if perm is not None:
            ids, sparse, q_sorted = ids[perm], sparse[perm], q_no[perm]
        else:
            q_sorted = q_no
        # which candidates do I hold?  (local look-up; unknown here = held elsewhere or nowhere)
        table = self.local._ids
        lookup = table._psg_ordinal if self.local.mode.value == 1 else table._doc_ordinal
        uniq, inv = np.unique(ids, return_inverse=True)
        ord_u = np.fromiter((lookup.get(u, -1) for u in uniq), dtype=np.int64, count=len(uniq))
        ordinals = ord_u[inv]
        # control traffic travels on the backend's kind of memory: device tensors under RCCL
        use_gpu = dist.get_backend(self.group) != "gloo"
        held = torch.from_numpy((ordinals >= 0).astype(np.int32))
        if use_gpu:
            held = held.cuda(self.device)
        total = held.clone()
        dist.all_reduce(total, group=self.group)  # every pair must be held by exactly one rank
        total_h = total.cpu().numpy()
        if total_h.size and int(total_h.min()) < 1:
            bad = int(np.flatnonzero(total_h < 1)[0])
            raise IndexError(f"ID {ids[bad]} not found in the index.")
        mine = np.flatnonzero(ordinals >= 0)
        local_off = np.zeros(n_q + 1, dtype=np.int64)
        np.cumsum(np.bincount(q_sorted[mine], minlength=n_q), out=local_off[1:])
        if mine.size:
            dense_local, _, _ = self.local._score_pairs(query_vectors, local_off, ordinals[mine].astype(np.int32), self.local.mode)
        else:
            dense_local = np.zeros(0, dtype=np.float32)
        # all-to-all to the owners of the queries (contiguous blocks of q_no)
        q_bounds = query_partition(n_q, self.world)
        send_counts = np.diff(local_off[q_bounds])
        holder = np.full(ids.shape[0], -1, dtype=np.int64)
        gathered = [torch.zeros_like(held) for _ in range(self.world)]
        dist.all_gather(gathered, held, group=self.group)
        for r, h in enumerate(gathered):
            holder[h.cpu().numpy() > 0] = r
        q_lo, q_hi = int(q_bounds[self.rank]), int(q_bounds[self.rank + 1])
        lo, hi = int(pair_off[q_lo]), int(pair_off[q_hi])
        own_holder = holder[lo:hi]
        recv_counts = np.bincount(own_holder, minlength=self.world).astype(np.int64)
        send = torch.from_numpy(dense_local)
        if use_gpu:
            send = send.cuda(self.device)
        recv = exchange(send, send_counts, recv_counts, self.group)
        recv = recv.cpu().numpy() if use_gpu else recv.numpy()
        arrival = np.argsort(own_holder, kind="stable")
        dense = np.empty(hi - lo, dtype=np.float32)
        dense[arrival] = recv
        # owner-side epilogue on the device: interpolation + order (ffx_aggregate with 1-row segments)
        from fast_forward_amd import native

        own_off = pair_off[q_lo:q_hi + 1] - lo
        seg = np.arange(hi - lo + 1, dtype=np.int64)
        if hi > lo:
            ff, interp, order = native.aggregate(dense, seg, 1, sparse[lo:hi] if alpha is not None else None,
                                                 0.0 if alpha is None else alpha, own_off, sort=sort, device=self.device)
        else:
            ff, interp, order = dense, (dense if alpha is not None else None), np.zeros(0, dtype=np.int32)
        rows = np.arange(lo, hi) if perm is None else perm[lo:hi]
        if sort:
            rows, score = rows[order], (interp if alpha is not None else ff)[order]
        else:
            score = interp if alpha is not None else ff
        out = df.iloc[rows].reset_index(drop=True)
        out["score"] = score.astype(df.dtypes["score"])
        return out

    def _finish(self, frame, ranking, name, gather: bool):
        import pandas as pd

        from fast_forward_amd.ranking import Ranking

        if gather:
            frames = [None] * self.world
            self._dist.all_gather_object(frames, frame, group=self.group)
            frame = pd.concat(frames, ignore_index=True)
        return Ranking(Ranking._trusted(frame), name=name, dtype=ranking._df.dtypes["score"], copy=False, is_sorted=False)

    def __call__(self, ranking, gather: bool = False):
        """Dense scores (``Index.__call__``) of the queries this rank owns, or of all with ``gather``."""
        return self._finish(self._score(ranking, None, sort=False), ranking, "fast-forward", gather)

    def rerank(self, ranking, alpha: float, gather: bool = False):
        """``ranking.interpolate(index(ranking), alpha)`` with the epilogue on the owners' devices."""
        return self._finish(self._score(ranking, alpha, sort=True), ranking, ranking.name, gather)

    def close(self) -> None:
        self.local.close()
