"""``InMemoryIndex``: the Fast-Forward index held in HBM, scored by the HIP kernels of libffx.

Drop-in for ``fast_forward.index.InMemoryIndex`` (reference ``src/fast_forward/index/memory.py:20-180``)
on the re-ranking path.  "Memory" here is the device's: vectors live in one contiguous fp32
matrix in HBM (288 GB per MI355X; 3 KB per 768-d row => ~93 M rows per GPU), ids live on the
host as ordinal tables, and ``Index.__call__`` runs

    ids -> ordinals (host)  ->  expand / gather_dot / segment-reduce / interpolate (device)

Differences from the reference, all deliberate:
  * storage is float32, or float16 when the first ``add`` hands over float16 vectors (the
    reference likewise keeps the dtype of the first chunk, memory.py:80-83; float32 queries
    times float16 rows are float32 products in NumPy, which is what the kernels compute).
    The reference silently switches to float64 for every chunk after the first
    (memory.py:106) and so returns float64-accumulated scores on large indexes; here scores
    always follow the semantics of a single-chunk reference index.
  * the vector store grows geometrically instead of by a list of chunks; ``init_size`` /
    ``alloc_size`` keep their meaning as the initial capacity and the minimum growth step.
  * no CPU fallback: without libffx.so or a HIP device, construction fails.
"""

from __future__ import annotations

import logging
from collections.abc import Iterable, Iterator

import numpy as np

from fast_forward_amd import native
from fast_forward_amd.encoder import Encoder
from fast_forward_amd.index.api import IDSequence, Index, Mode
from fast_forward_amd.index.ids import IdTable

LOGGER = logging.getLogger(__name__)


class InMemoryIndex(Index):
    """Fast-Forward index resident in HBM."""

    def __init__(self, query_encoder: Encoder | None = None, quantizer=None, mode: Mode = Mode.MAXP,
                 encoder_batch_size: int = 32, init_size: int = 2**16, alloc_size: int = 2**16,
                 device: int = 0, exact_sum: bool = True, dtype=None) -> None:
        """
        :param dtype: row storage, ``numpy.float32`` or ``numpy.float16``; ``None`` (default)
            follows the first ``add``: float16 vectors are stored as float16, anything else as
            float32.  Float16 halves the bytes per row (twice the rows per GPU, ~1.9x the
            queries/s) at the cost of rounding the embeddings once.
        :param init_size: initial capacity in vectors (allocated on the first ``add``).
        :param alloc_size: minimum growth step in vectors.
        :param device: HIP device ordinal holding the index.
        :param exact_sum: ``True`` (default): scores are bit-identical to the reference's
            ``np.sum(q * d, axis=1)``, so run orderings are identical too.  ``False``: FMA
            summation, within 1e-4 of the reference on unit-scale vectors.
        """
        native.lib()  # fail now, loudly, if the native library is missing
        self._ids = IdTable()
        self._store: native.DeviceIndex | None = None
        self._plan: native.Plan | None = None
        self._n = 0
        self._init_size, self._alloc_size = int(init_size), int(alloc_size)
        self._device = int(device)
        self._dtype = None if dtype is None else np.dtype(dtype)
        self._ids_on_device = -1
        self._owns_store = True
        self._rotation: np.ndarray | None = None    # OPQ: queries are rotated instead of rows
        self._host_codes: list[np.ndarray] = []     # codes of a quantizer the device cannot read
        self.exact_sum = bool(exact_sum)
        self.last_stats: dict | None = None
        self.last_placement: dict | None = None     # what consolidate() measured (ffx_index_reseat)
        super().__init__(query_encoder=query_encoder, quantizer=quantizer, mode=mode, encoder_batch_size=encoder_batch_size)

    @classmethod
    def adopt(cls, store: "native.DeviceIndex", doc_ids=None, psg_ids=None, doc_off=None, n_rows: int | None = None,
              query_encoder: Encoder | None = None, mode: Mode = Mode.MAXP, exact_sum: bool = True) -> "InMemoryIndex":
        """An ``InMemoryIndex`` over rows that are ALREADY resident in a ``DeviceIndex`` (filled on the
        device, uploaded through the C ABI, or moved there from a file): only the ids are registered.
        ``doc_ids`` / ``psg_ids`` name the rows one by one; or, with ``doc_off``, ``doc_ids[o]`` names
        the document owning rows ``doc_off[o] .. doc_off[o+1]``.  The store stays the caller's
        (``release`` detaches it; ``close`` does not free it)."""
        n = int(store.capacity if n_rows is None else n_rows)
        index = cls(query_encoder, mode=mode, device=store.device, exact_sum=exact_sum, dtype=store.dtype)
        if doc_ids is not None and doc_off is not None:
            doc_off = np.asarray(doc_off, dtype=np.int64)
            names = np.asarray(doc_ids)
            if (np.diff(doc_off) != 1).any():
                names = names[np.repeat(np.arange(doc_off.shape[0] - 1), np.diff(doc_off))]
            doc_ids = names
        if doc_ids is None and psg_ids is None:
            raise ValueError("Vector has neither document nor passage ID.")
        index._ids.add([None] * n if doc_ids is None else doc_ids, [None] * n if psg_ids is None else psg_ids, 0)
        index._store, index._n, index._dtype, index._owns_store = store, n, store.dtype, False
        return index

    def release(self) -> "native.DeviceIndex | None":
        """Detach the device store (its plans are closed) and hand it back; the index is empty afterwards."""
        store = self._store
        if self._plan is not None:
            self._plan.close()
            self._plan = None
        self._store, self._n = None, 0
        return store

    # ---- storage --------------------------------------------------------------------------------
    def _get_num_vectors(self) -> int:
        return self._n

    def _get_internal_dim(self) -> int | None:
        """Length of a stored row: the vector dimensionality, or the code length if quantized."""
        if self._store is None:
            return None
        return self._quantizer.dims[1] if self._quantizer is not None else self._store.dim

    def _stored(self, rows: np.ndarray) -> np.ndarray:
        """Stored representation of rows: vectors, or codes if a quantizer is attached."""
        if self._host_codes:
            return np.concatenate(self._host_codes)[rows]
        return self._store.gather(rows)

    def _get_doc_ids(self) -> set[str]:
        return self._ids.doc_ids

    def _get_psg_ids(self) -> set[str]:
        return self._ids.psg_ids

    def _reserve(self, n_total: int, dim: int) -> None:
        if self._store is not None and n_total <= self._store.capacity:
            return
        if self._store is None:
            cap = max(self._init_size, n_total)
        else:
            cap = max(n_total, self._store.capacity + self._alloc_size, int(self._store.capacity * 1.5))
        LOGGER.debug("growing the device vector store to %s rows", cap)
        grown = native.DeviceIndex(cap, dim, self._device, dtype=self._dtype)
        if self._store is not None:
            if self._store.codebook_shape is not None:
                grown.set_codebook(self._quantizer.device_codebook()[0])
            grown.copy_rows_from(self._store, self._n)
            if self._plan is not None:
                self._plan.close()
                self._plan = None
            if self._owns_store:
                self._store.close()
        self._store, self._owns_store = grown, True
        self._ids_on_device = -1

    def _add(self, vectors: np.ndarray, doc_ids: IDSequence, psg_ids: IDSequence, _uploader=None) -> None:
        """``vectors`` are codes when a quantizer is attached (``Index.add`` encodes first).

        Quantized rows are kept in HBM as uint8 product-quantizer codes when the quantizer exposes
        a codebook the kernels can read (``Quantizer.device_codebook``); for any other quantizer
        the codes stay on the host and their decoded vectors are what the device stores -- scoring
        decoded rows is exactly what the reference does (index/base.py:292-293)."""
        n_new = vectors.shape[0]
        codes = None
        if self._quantizer is not None:
            book = self._quantizer.device_codebook()
            if book is None:
                codes, vectors = vectors, np.ascontiguousarray(self._quantizer.decode(vectors), dtype=np.float32)
                self._dtype = np.dtype(np.float32)
            else:
                self._dtype = np.dtype(np.uint8)
        elif self._dtype is None:
            self._dtype = np.dtype(np.float16) if vectors.dtype == np.float16 else np.dtype(np.float32)
        fresh = self._store is None
        self._reserve(self._n + n_new, vectors.shape[1])
        if (fresh or self._store.codebook_shape is None) and self._dtype == np.uint8:
            codewords, self._rotation = self._quantizer.device_codebook()
            self._store.set_codebook(codewords)
        self._ids.add(doc_ids, psg_ids, self._n)  # a duplicate passage id raises before anything is modified
        if _uploader is not None:
            # rows that already sit in files (OnDiskIndex.to_memory): `vectors` only carries shape
            # and dtype, the rows go from the chunk maps to HBM natively (ffx_index_upload_rows)
            assert np.dtype(vectors.dtype) == self._dtype, "stored rows must already be in the index dtype"
            _uploader(self._store, self._n)
        else:
            self._store.upload(np.ascontiguousarray(vectors, dtype=self._dtype), self._n)
        if codes is not None:
            self._host_codes.append(np.array(codes, copy=True))
        self._n += n_new

    def consolidate(self, placement_tries: int = -1) -> None:
        """The reference merges its chunks into one array here (index/memory.py:121-131).  The device store is one matrix
        already; what is left of "I am done adding, lay the vectors out for look-ups" is WHERE that matrix lies in HBM:
        ``ffx_index_reseat`` moves it into the fastest of up to ``placement_tries`` further allocations (how fast a random
        gather runs belongs to the allocation: copies of one 27 GB index score the same batch 3 - 6 % apart).  Stores below
        256 MiB stay where they are; transient memory: ``placement_tries`` x the store (-1, the default: as many as half of the
        free memory holds, at most 7); ``last_placement`` says what was measured.  ``placement_tries=0``: nothing happens (the
        reference's call has no effect on results either)."""
        if self._store is not None and placement_tries != 0:
            self.last_placement = self._store.reseat(placement_tries)

    def _sync_ids(self) -> None:
        if self._store is not None and self._ids_on_device != self._ids.version:
            doc_off, doc_rows = self._ids.doc_csr()
            self._store.set_doc_rows(doc_off, doc_rows)
            self._store.set_psg_rows(self._ids.psg_rows())
            self._ids_on_device = self._ids.version

    def _get_vectors(self, ids: Iterable[str]) -> tuple[np.ndarray, list[str]]:
        rows, owners = self._ids.rows_for(ids, self.mode.value)
        if not rows:
            return np.array([]), []
        return self._stored(np.asarray(rows, dtype=np.int32)), owners

    def _batch_iter(self, batch_size: int) -> Iterator[tuple[np.ndarray, IDSequence, IDSequence]]:
        docs, psgs = self._ids.row_ids(self._n)
        for lo in range(0, self._n, batch_size):
            hi = min(lo + batch_size, self._n)
            yield self._stored(np.arange(lo, hi, dtype=np.int32)), docs[lo:hi], psgs[lo:hi]

    # ---- scoring ----------------------------------------------------------------------------------
    def _resolve_ids(self, ids: np.ndarray, mode: Mode) -> np.ndarray:
        return self._ids.resolve(ids, mode.value)

    def _result_buffers(self, n: int, wanted=(True, True, True)):
        """Page-locked (ff, interp, order) arrays of at least ``n`` elements, kept with the index: results land in
        them at link rate and without a page fault per 4 KB (a fresh 140 MB array costs ~10 ms just to touch).
        Only the ``wanted`` ones are made (page-locking 157 MB takes ~40 ms); the others come back as None.
        Only for callers that consume the results before the next call (the coded paths of ``__call__`` /
        ``rerank`` gather out of them at once)."""
        have = getattr(self, "_out_cache", None)
        if have is None:
            have = self._out_cache = [None, None, None]
        for i, dtype in enumerate((np.float32, np.float32, np.int32)):
            if wanted[i] and (have[i] is None or have[i].size < n):
                have[i] = None  # (the old block goes first)
                have[i] = native.pinned_empty(n + n // 8, dtype)
        return tuple(a if w else None for a, w in zip(have, wanted))

    def _score_pairs(self, query_vectors, pair_off, pair_ids, mode: Mode, sparse=None, alpha=None, sort=False, want_dense=True,
                     reuse_outputs=False):
        if self._store is None:
            raise IndexError("the index is empty")
        q = np.ascontiguousarray(query_vectors, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self._store.query_dim:
            raise ValueError(f"Query vector dimensionality ({q.shape[-1]}) does not match index dimensionality ({self._store.query_dim}).")
        if self._rotation is not None:
            # OPQ decodes to c @ R.T, and <q, c R^T> = <q R, c>: rotate the Q queries once instead
            # of every gathered row (equal up to fp32 rounding of the rotation, not bit-identical)
            q = np.ascontiguousarray(q @ self._rotation, dtype=np.float32)
        self._sync_ids()
        n_pairs = int(pair_off[-1])
        out = (self._result_buffers(n_pairs, (want_dense or sparse is None, sparse is not None, bool(sort)))
               if (reuse_outputs and n_pairs >= (1 << 20)) else None)
        if n_pairs >= self._PIPELINED_FROM:
            # large batches go through ffx_score: whole-query slices rotate through three plans, so the upload of
            # one slice, the kernels of the one before and the download of the one before that overlap
            # (18.5 ms instead of ~31 for 34.9 M pairs); results are the same bits
            ff, interp, order, self.last_stats = native.score(
                self._store, q, pair_off, pair_ids, mode.value, sparse, 0.0 if alpha is None else alpha, exact_sum=self.exact_sum,
                sort=sort, out=out, want_ff=want_dense or sparse is None)
            return ff, interp, order
        if self._plan is None:
            self._plan = native.Plan(self._store)
        plan = self._plan
        plan.set_queries(q)
        plan.set_pairs(pair_off, pair_ids, sparse, mode.value)
        plan.run(0.0 if alpha is None else alpha, exact_sum=self.exact_sum, sort=sort)
        ff, interp, order = plan.fetch(ff=want_dense or sparse is None, interp=sparse is not None, order=sort, out=out)
        self.last_stats = plan.stats()
        return ff, interp, order

    def _score_ranked(self, query_vectors, pair_off, pair_ids, mode: Mode, sparse, alpha, carry, want_order: bool = False, keep: bool = False):
        """The run in final order straight off the device (``ffx_score_ranked``): (scores, carried codes, order | None).
        Scores are the interpolated ones when ``sparse`` is given, else the dense ones.  ``keep``: the dense scores and the
        carried codes stay in HBM (8 bytes per candidate) for a ``Ranking.interpolate`` that follows (``_rank_kept``)."""
        if self._store is None:
            raise IndexError("the index is empty")
        q = np.ascontiguousarray(query_vectors, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self._store.query_dim:
            raise ValueError(f"Query vector dimensionality ({q.shape[-1]}) does not match index dimensionality ({self._store.query_dim}).")
        if self._rotation is not None:
            q = np.ascontiguousarray(q @ self._rotation, dtype=np.float32)
        self._sync_ids()
        scores, codes, order, self.last_stats = native.score_ranked(self._store, q, pair_off, pair_ids, mode.value, sparse, 0.0 if alpha is None else alpha,
                                                                    carry=carry, exact_sum=self.exact_sum, want_order=want_order, keep=keep)
        return scores, codes, order

    def _kept_token(self) -> int:
        """Token of the dense scores the device keeps from the last ``__call__`` (0: none)."""
        return 0 if self._store is None else native.kept_token(self._store)

    def _rank_kept(self, token: int, pair_off, sparse, alpha: float):
        """``ffx_rank_kept``: (interpolated scores, id codes) in final order from the dense scores kept under ``token``."""
        scores, codes, _, self.last_stats = native.rank_kept(self._store, token, pair_off, sparse, alpha)
        return scores, codes

    _PIPELINED_FROM = 6 << 20  # pairs: below this ffx_score would run a single slice anyway

    def _warm_up(self, n_queries: int, n_rows: int, has_sparse: bool, result_blocks: list):
        """Beside the id look-up of a first call, on the helper thread: the id structure goes to the device if it is not
        there yet (``_sync_ids``: 0.1 s for 8.8 M documents), the plans / streams / device buffers ``ffx_score_ranked`` would
        otherwise allocate on its way are made (``ffx_index_prepare_score``), then the result blocks are page-locked."""
        def device_side():
            if self._store is not None:
                self._sync_ids()
                self._store.prepare_score(n_queries, n_rows, has_sparse, True, False)

        return native.reserve_pooled(result_blocks, first=device_side)

    def _early_stopping_kept(self, id_at, sparse, pair_off, query_vectors, cutoff: int, alpha: float, depths, ordinals):
        """Early stopping (base.py:316-387) resident on the device (``ffx_plan_early_stopping``): the active mask, the
        running best dense score and the running top-``cutoff`` threshold of every query stay in HBM between depth
        intervals; one upload of the candidate lists (unknown ids as -1: an ``IndexError`` only if some interval reaches
        one, as in the reference), and only the candidates that got scored come back, compacted on the device
        (``ffx_plan_fetch_scored``).  Float64 first-stage scores take the NumPy statement (``Index._early_stopping_kept``)."""
        sparse = np.asarray(sparse)
        if self._store is None:
            raise IndexError("the index is empty")
        if sparse.dtype != np.float32:
            return super()._early_stopping_kept(id_at, sparse, pair_off, query_vectors, cutoff, alpha, depths, ordinals)
        q = np.ascontiguousarray(query_vectors, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self._store.query_dim:
            raise ValueError(f"Query vector dimensionality ({q.shape[-1]}) does not match index dimensionality ({self._store.query_dim}).")
        if self._rotation is not None:
            q = np.ascontiguousarray(q @ self._rotation, dtype=np.float32)
        self._sync_ids()
        if self._plan is None:
            self._plan = native.Plan(self._store)
        plan = self._plan
        plan.set_queries(q)
        plan.set_pairs_lazy(pair_off, ordinals, sparse, self.mode.value)
        try:
            scored, intervals = plan.early_stopping(cutoff, alpha, depths, exact_sum=self.exact_sum, bound_f32=self._es_bound_f32())
        except native.NotFound as e:
            raise IndexError(f"ID {id_at(e.pair)} not found in the index.") from None
        self.last_stats = dict(plan.stats(), early_stopping_scored_pairs=scored, early_stopping_intervals=intervals)
        LOGGER.info("early stopping: %s of %s candidates scored in %s depth intervals", scored, len(sparse), intervals)
        return plan.fetch_scored(scored)

    def _native_id_table(self, mode: Mode):
        return self._ids.native_table(mode.value)

    def close(self) -> None:
        """Release device memory now (otherwise released when the object is collected)."""
        if self._plan is not None:
            self._plan.close()
            self._plan = None
        if self._store is not None:
            if self._owns_store:
                self._store.close()
            self._store = None
        self._out_cache = None
        self._n = 0
        try:  # idle page-locked result blocks (up to FFX_RESULT_POOL_BYTES of host memory) go with the index that filled them
            native.drain_result_pool(wait=False)
        except Exception:  # noqa: BLE001
            pass

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:
            pass
