"""``Mode`` and the abstract ``Index``: the re-ranking entry point ``Index.__call__``.

Mirrors ``fast_forward.index.Index`` (reference ``src/fast_forward/index/base.py:18-513``) for
the re-ranking path: same constructor and ``__call__`` arguments, same exceptions.  Where the
reference computes scores with NumPy inside pandas merges (``_compute_scores``,
base.py:279-314), this class hands integer candidate lists to a device backend
(``_score_pairs``), implemented by ``InMemoryIndex`` on the HIP kernels of libffx.
"""

from __future__ import annotations

import abc
import logging
import weakref
from collections.abc import Iterable, Iterator, Sequence
from enum import Enum
from time import perf_counter

import numpy as np
import pandas as pd

from fast_forward_amd.encoder import Encoder
from fast_forward_amd.ranking import Ranking

LOGGER = logging.getLogger(__name__)

IDSequence = Sequence["str | None"]


# what pandas.read_csv turns into NaN by default (Ranking.from_file, ranking.py:400-409)
_PANDAS_NA_STRINGS = frozenset(["", "#N/A", "#N/A N/A", "#NA", "-1.#IND", "-1.#QNAN", "-NaN", "-nan", "1.#IND", "1.#QNAN", "<NA>",
                                "N/A", "NA", "NULL", "NaN", "None", "n/a", "nan", "null"])


class Mode(Enum):
    """Ranking mode of an index (values as in the reference, base.py:18-24)."""

    PASSAGE = 1
    MAXP = 2
    FIRSTP = 3
    AVEP = 4


_CODED_MIN_ROWS = 4096  # from here on results are returned in coded form (the frame is built on first use)


def _in_final_order(q_order: Sequence[str]) -> bool:
    """Query ids strictly descending: the order a ``Ranking`` the constructor sorted has
    (ranking.py:115-117).  A user-built ``Ranking(..., is_sorted=True)`` need not be."""
    return all(q_order[i] > q_order[i + 1] for i in range(len(q_order) - 1))


def _grouped_by_query(q_no: np.ndarray, n_queries: int):
    """CSR over pairs: a permutation that makes every query's pairs contiguous (identity when
    they already are) and the offsets of each query."""
    counts = np.bincount(q_no, minlength=n_queries).astype(np.int64)
    pair_off = np.zeros(n_queries + 1, dtype=np.int64)
    np.cumsum(counts, out=pair_off[1:])
    if (np.diff(q_no) >= 0).all():
        return None, pair_off
    return np.argsort(q_no, kind="stable"), pair_off


class Index(abc.ABC):
    """Abstract Fast-Forward index."""

    _query_encoder: Encoder | None = None
    _quantizer = None

    #: How early stopping evaluates ``alpha * s_last + (1 - alpha) * max_dense`` (base.py:353-355: a Python float times a
    #: ``np.float32`` SCALAR, so the reference's own result depends on NumPy's scalar promotion).  ``"numpy1"`` (default):
    #: in float64, as under the reference's pinned ``numpy>=1.24,<2`` (pyproject.toml:16); ``"numpy2"``: in float32, as
    #: under NumPy >= 2 (NEP 50).  Both are pinned by fixtures the reference produced under either NumPy
    #: (tests/golden/early_stopping_np1.npz, early_stopping_np2.npz); a flipped decision changes which rows exist.
    early_stopping_promotion: str = "numpy1"

    def _es_bound_f32(self) -> bool:
        if self.early_stopping_promotion not in ("numpy1", "numpy2"):
            raise ValueError('early_stopping_promotion must be "numpy1" or "numpy2"')
        return self.early_stopping_promotion == "numpy2"

    def __init__(self, query_encoder: Encoder | None = None, quantizer=None, mode: Mode = Mode.MAXP,
                 encoder_batch_size: int = 32) -> None:
        super().__init__()
        if query_encoder is not None:
            self.query_encoder = query_encoder
        self.mode = mode
        if quantizer is not None:
            self.quantizer = quantizer
        self._encoder_batch_size = encoder_batch_size

    # ---- configuration ---------------------------------------------------------------------
    @property
    def query_encoder(self) -> Encoder | None:
        return self._query_encoder

    @query_encoder.setter
    def query_encoder(self, encoder: Encoder) -> None:
        assert isinstance(encoder, Encoder)
        self._query_encoder = encoder

    @property
    def quantizer(self):
        return self._quantizer

    def _on_quantizer_set(self) -> None:
        """Hook for subclasses (the reference's OnDiskIndex stores the quantizer here)."""

    @quantizer.setter
    def quantizer(self, quantizer) -> None:
        """Attach a trained quantizer; only possible while the index is empty (base.py:104-118)."""
        from fast_forward_amd.quantizer import Quantizer

        assert isinstance(quantizer, Quantizer)
        if len(self) > 0:
            raise RuntimeError("Quantizers can only be attached to empty indexes.")
        self._quantizer = quantizer
        self._on_quantizer_set()
        quantizer.set_attached()

    @property
    def mode(self) -> Mode:
        return self._mode

    @mode.setter
    def mode(self, mode: Mode) -> None:
        assert isinstance(mode, Mode)
        self._mode = mode

    def encode_queries(self, queries: Sequence[str]) -> np.ndarray:
        """Encode query texts in batches of ``encoder_batch_size`` (base.py:58-72)."""
        if self.query_encoder is None:
            raise RuntimeError("Index does not have a query encoder.")
        step = self._encoder_batch_size
        return np.concatenate([self.query_encoder(queries[i:i + step]) for i in range(0, len(queries), step)])

    # ---- storage interface (implemented by subclasses) ------------------------------------------
    @abc.abstractmethod
    def _get_internal_dim(self) -> int | None: ...

    @abc.abstractmethod
    def _get_doc_ids(self) -> set[str]: ...

    @abc.abstractmethod
    def _get_psg_ids(self) -> set[str]: ...

    @abc.abstractmethod
    def _get_num_vectors(self) -> int: ...

    @abc.abstractmethod
    def _add(self, vectors: np.ndarray, doc_ids: IDSequence, psg_ids: IDSequence) -> None: ...

    @abc.abstractmethod
    def _get_vectors(self, ids: Iterable[str]) -> tuple[np.ndarray, list[str]]:
        """Vectors needed to score ``ids`` in the current mode, and the id owning each vector;
        ``IndexError`` for an unknown id (base.py:258-277)."""

    @abc.abstractmethod
    def _batch_iter(self, batch_size: int) -> Iterator[tuple[np.ndarray, IDSequence, IDSequence]]:
        """Stored vectors (codes, if a quantizer is attached) with their ids."""

    @abc.abstractmethod
    def _resolve_ids(self, ids: np.ndarray, mode: Mode) -> np.ndarray:
        """String ids -> int32 ordinals; ``IndexError`` for an unknown id."""

    @abc.abstractmethod
    def _score_pairs(self, query_vectors: np.ndarray, pair_off: np.ndarray, pair_ids: np.ndarray, mode: Mode,
                     sparse: np.ndarray | None = None, alpha: float | None = None,
                     sort: bool = False, want_dense: bool = True) -> tuple[np.ndarray, np.ndarray | None, np.ndarray | None]:
        """Device scoring of CSR candidate lists: ``(dense, interpolated|None, order|None)``."""

    @property
    def dim(self) -> int | None:
        """Dimensionality of the vectors the index takes (before quantization, if any)."""
        if self._quantizer is not None:
            return self._quantizer.dims[0]
        return self._get_internal_dim()

    @property
    def doc_ids(self) -> set[str]:
        return self._get_doc_ids()

    @property
    def psg_ids(self) -> set[str]:
        return self._get_psg_ids()

    def __len__(self) -> int:
        return self._get_num_vectors()

    def add(self, vectors: np.ndarray, doc_ids: IDSequence | None = None, psg_ids: IDSequence | None = None) -> None:
        """Add vectors with their document and/or passage ids (validation of base.py:211-256).

        ``ValueError``: id count mismatch, wrong dimensionality, a vector without any id.
        ``RuntimeError``: a passage id that already exists."""
        n, dim = vectors.shape
        doc_ids = [None] * n if doc_ids is None else doc_ids
        psg_ids = [None] * n if psg_ids is None else psg_ids
        if not len(doc_ids) == len(psg_ids) == n:
            raise ValueError("Number of IDs does not match number of vectors.")
        if self.dim is not None and dim != self.dim:
            raise ValueError(f"Input vector dimensionality ({dim}) does not match index dimensionality ({self.dim}).")
        if any(d is None and p is None for d, p in zip(doc_ids, psg_ids)):
            raise ValueError("Vector has neither document nor passage ID.")
        self._add(vectors if self._quantizer is None else self._quantizer.encode(vectors), doc_ids, psg_ids)

    # ---- scoring -------------------------------------------------------------------------------------
    def _numbered_queries(self, ranking: Ranking) -> tuple[np.ndarray, list[str], list[str]]:
        """``q_no`` per row = order of first appearance of the query (base.py:418-421), plus the
        query ids and texts in that order."""
        q_no, q_ids = ranking._numbering()
        return q_no.astype(np.int64, copy=False), list(q_ids), ranking._query_texts()

    def _compute_scores(self, data: pd.DataFrame, query_vectors: np.ndarray) -> pd.DataFrame:
        """``data`` (columns ``id``, ``q_no``, ...) plus an ``ff_score`` column: the dense score of
        every row in the current mode (the contract of base.py:279-314, computed on the device)."""
        out = data.copy()
        if len(out) == 0:
            out["ff_score"] = np.zeros(0, dtype=np.float32)
            return out
        q_no = out["q_no"].to_numpy(dtype=np.int64)
        # __call__ resolves the whole ranking once and passes the ordinals along
        ordinals = out.pop("_ord").to_numpy(dtype=np.int32) if "_ord" in out.columns else \
            self._resolve_ids(out["id"].to_numpy(), self.mode)
        perm, pair_off = _grouped_by_query(q_no, query_vectors.shape[0])
        dense, _, _ = self._score_pairs(query_vectors, pair_off, ordinals if perm is None else ordinals[perm], self.mode)
        if perm is not None:
            unperm = np.empty_like(dense)
            unperm[perm] = dense
            dense = unperm
        out["ff_score"] = dense
        return out

    def _early_stopping(self, df: pd.DataFrame, query_vectors: np.ndarray, cutoff: int, alpha: float,
                        depths: Iterable[int]) -> pd.DataFrame:
        """Frame form of :meth:`_early_stopping_scores` (columns ``id``, ``q_no``, ``score`` in,
        plus ``ff_score`` out; NaN where a candidate was not scored)."""
        out = df.drop(columns=["_ord"], errors="ignore")
        out["ff_score"] = self._early_stopping_scores(df["id"].to_numpy(), df["q_no"].to_numpy(dtype=np.int64), df["score"].to_numpy(),
                                                      query_vectors, cutoff, alpha, depths)
        return out

    def _early_stopping_scores(self, ids: np.ndarray, q_no: np.ndarray, sparse: np.ndarray, query_vectors: np.ndarray,
                               cutoff: int, alpha: float, depths: Iterable[int]) -> np.ndarray:
        """Score each query only as deep as needed for its top-``cutoff`` (base.py:316-387).

        Depth intervals ``[a, b)`` over the sorted depths ``>= cutoff``; after the first
        interval a query continues only while its ``cutoff``-th best interpolated score is below
        the best it could still reach, ``alpha * s_last + (1 - alpha) * max_dense`` (``s_last``:
        sparse score of the last scored candidate).  Returns the dense scores in row order, NaN
        for candidates that were never scored.  Ids are resolved interval by interval, so -- as in
        the reference -- an id that is never reached is never looked up."""
        n, n_q = q_no.shape[0], query_vectors.shape[0]
        perm, pair_off = _grouped_by_query(q_no, n_q)
        if perm is None:
            rank_in_q = np.arange(n, dtype=np.int64) - pair_off[q_no]
        else:
            rank_in_q = pd.Series(q_no).groupby(q_no, sort=False).cumcount().to_numpy()
        # dense scores are float32 (as the reference's np.sum over float32 vectors); the criterion below is the
        # reference's SCALAR expression `python_float * np.float32` under the chosen NumPy promotion
        # (early_stopping_promotion): float64 products and sum under numpy < 2, float32 throughout under numpy >= 2
        bound_f32 = self._es_bound_f32()
        ff = np.full(n, np.nan, dtype=np.float32)
        interp = np.full(n, np.nan, dtype=np.result_type(sparse.dtype, np.float32))
        active = np.ones(n_q, dtype=bool)
        a = 0
        for b in sorted(depths):
            if b < cutoff:
                continue
            if a > 0:
                # continue-criterion for all queries at once.  Scored rows, grouped by query:
                rows = np.flatnonzero(~np.isnan(ff))
                rows = rows[np.argsort(q_no[rows], kind="stable")]          # frame order kept inside a query
                qs = q_no[rows]
                starts = np.flatnonzero(np.r_[True, qs[1:] != qs[:-1]])
                ends = np.r_[starts[1:], rows.size]
                present = qs[starts]
                # cutoff-th best interpolated score so far (the worst one if fewer were scored)
                by_score = rows[np.lexsort((-interp[rows].astype(np.float64), qs))]
                kth = interp[by_score[np.minimum(starts + cutoff, ends) - 1]]
                # best it could still reach: alpha * (sparse score of the last scored candidate) +
                # (1 - alpha) * (best dense score so far) -- NumPy scalar promotion as in the reference
                last_sparse = sparse[rows[ends - 1]]
                best_dense = np.maximum.reduceat(ff[rows], starts)
                if bound_f32:  # (arrays times a Python float keep their dtype under either NumPy)
                    bound = alpha * last_sparse + (1 - alpha) * best_dense
                else:
                    bound = np.float64(alpha) * last_sparse.astype(np.float64) + np.float64(1 - alpha) * best_dense.astype(np.float64)
                keep = np.zeros(n_q, dtype=bool)
                keep[present] = kth < bound
                active &= keep
            LOGGER.info("depth %s: %s queries left", b, int(active.sum()))
            take = np.flatnonzero(active[q_no] & (rank_in_q >= a) & (rank_in_q < b))
            if take.size == 0:
                break
            part_perm, part_off = _grouped_by_query(q_no[take], n_q)
            if part_perm is not None:
                take = take[part_perm]
            chunk_ff, _, _ = self._score_pairs(query_vectors, part_off, self._resolve_ids(ids[take], self.mode), self.mode)
            ff[take] = chunk_ff
            interp[take] = alpha * sparse[take] + (1 - alpha) * chunk_ff
            a = b
        return ff

    def __call__(self, ranking: Ranking, early_stopping: int | None = None, early_stopping_alpha: float | None = None,
                 early_stopping_depths: Iterable[int] | None = None, batch_size: int | None = None) -> Ranking:
        """Dense scores for every (query, candidate) of ``ranking`` (base.py:389-469).

        ``ValueError``: ranking without queries; early stopping without alpha/depths.
        ``IndexError``: a candidate id that is not in the index (raised before any device work).
        """
        if not ranking.has_queries:
            raise ValueError("Input ranking has no queries attached.")
        if early_stopping is not None and (early_stopping_alpha is None or early_stopping_depths is None):
            raise ValueError("Early stopping requires alpha and depths.")
        t0 = perf_counter()

        q_no, q_order, texts = self._numbered_queries(ranking)
        query_vectors = np.ascontiguousarray(self.encode_queries(texts))
        n_q = len(texts)
        grouped, block_off = ranking._csr() if len(q_no) > 0 else (0, None)
        final_order = _in_final_order(q_order)
        if (early_stopping is None and grouped is None and final_order and len(q_no) >= _CODED_MIN_ROWS
                and (batch_size is None or batch_size >= n_q)):
            # a large run in the constructor's order: integer codes in, integer codes out -- no Python string
            # per row on the way to the device, none on the way back (the frame is built on first use)
            codes, names = ranking._id_coding()
            dense, _, order = self._score_pairs(query_vectors, block_off, self._ordinals_of_codes(ranking, codes, names), self.mode, sort=True,
                                                **self._reuse_kw())
            from fast_forward_amd import native

            # `order` is a permutation inside every query: the query numbers stay as they are
            LOGGER.info("computed scores in %s seconds", perf_counter() - t0)
            out = Ranking._assemble_coded("fast-forward", q_no, q_order, native.take32(codes, order), names,
                                          native.take32(dense, order).astype(ranking._score_dtype(), copy=False), texts)
            # row i of the result is row order[i] of `ranking`: ranking.interpolate(result, alpha) needs no join
            if order.base is not None and getattr(self, "_result_buffers", None) is not None:
                # `order` is a view of the index's re-used result block: the result borrows it until the index needs the
                # block again (then it gets its own copy, InMemoryIndex._result_buffers) -- the interpolate that usually
                # follows at once reads it in place
                out._source_rows = (weakref.ref(ranking), order)
                self._order_borrower = weakref.ref(out)
            else:
                out._source_rows = (weakref.ref(ranking), order.copy())
            return out
        if (early_stopping is not None and final_order and grouped is None and len(q_no) >= _CODED_MIN_ROWS and (batch_size is None or batch_size >= n_q)
                and getattr(self, "_early_stopping_takes_ordinals", False) and ranking._score_dtype() == np.float32):
            # a large run, early stopping resident on the device: codes in, codes out (no frame, no Python string per row)
            from fast_forward_amd import native

            codes, names = ranking._id_coding()
            lenient = self._ordinals_of_codes(ranking, codes, names, missing_ok=True)
            if hasattr(self, "_early_stopping_scored"):  # only the scored candidates come back, compacted on the device
                rows, kept_scores = self._early_stopping_scored(lambda p: names[codes[p]], q_no, ranking._scores(), query_vectors, early_stopping,
                                                                early_stopping_alpha, early_stopping_depths, lenient, (None, block_off))
            else:
                ff = self._early_stopping_scores(lambda p: names[codes[p]], q_no, ranking._scores(), query_vectors, early_stopping, early_stopping_alpha,
                                                 early_stopping_depths, ordinals=lenient, grouped=(None, block_off))
                rows = np.flatnonzero(~np.isnan(ff)).astype(np.int32)  # the candidates that were scored, still grouped by query
                kept_scores = native.take32(np.ascontiguousarray(ff), rows)
            q_kept = np.asarray(q_no)[rows]
            off = np.zeros(n_q + 1, dtype=np.int64)
            np.cumsum(np.bincount(q_kept, minlength=n_q), out=off[1:])
            order = native.host_order(kept_scores, off)  # score descending inside every query, ties by first-stage rank
            src = native.take32(rows, order)
            present = np.flatnonzero(np.diff(off))
            renumber = np.full(n_q, -1, dtype=np.int64)
            renumber[present] = np.arange(present.shape[0])
            LOGGER.info("computed scores in %s seconds", perf_counter() - t0)
            return Ranking._assemble_coded("fast-forward", renumber[q_kept], [q_order[i] for i in present], native.take32(codes, src), names,
                                           native.take32(kept_scores, order).astype(ranking._score_dtype(), copy=False),
                                           None if texts is None else [texts[i] for i in present])
        df = ranking._df
        if early_stopping is not None and len(df) > 0 and final_order and (batch_size is None or batch_size >= n_q or grouped is None):
            # arrays in, arrays out: only the candidates that get scored are looked up and kept
            ids_col, sparse_col = df["id"].to_numpy(), df["score"].to_numpy()
            lenient = None
            if len(df) >= _CODED_MIN_ROWS and getattr(self, "_early_stopping_takes_ordinals", False):
                # a large run: its distinct ids are looked up once (unknown -> -1: an error only if reached)
                lenient = self._ordinals_of_codes(ranking, *ranking._id_coding(), missing_ok=True)
            extra = (lambda lo, hi: {}) if lenient is None else (lambda lo, hi: {"ordinals": lenient[lo:hi]})
            if batch_size is None or batch_size >= n_q:
                ff = self._early_stopping_scores(ids_col, q_no, sparse_col, query_vectors, early_stopping, early_stopping_alpha,
                                                 early_stopping_depths, **extra(0, len(df)))
            else:  # `batch_size` queries at a time
                ff = np.full(len(df), np.nan, dtype=np.float32)
                for qa in range(0, n_q, batch_size):
                    pa, pb = int(block_off[qa]), int(block_off[min(n_q, qa + batch_size)])
                    if pb > pa:
                        ff[pa:pb] = self._early_stopping_scores(ids_col[pa:pb], q_no[pa:pb], sparse_col[pa:pb], query_vectors,
                                                                early_stopping, early_stopping_alpha, early_stopping_depths, **extra(pa, pb))
            rows = np.flatnonzero(~np.isnan(ff))
            rows = rows[np.lexsort((-ff[rows].astype(np.float64), q_no[rows]))]  # q_no ascending = q_id descending
            cols = {c: df[c].to_numpy()[rows] for c in ("q_id", "id")}
            cols["score"] = ff[rows].astype(df.dtypes["score"], copy=False)
            cols["query"] = df["query"].to_numpy()[rows]
            present = np.flatnonzero(np.bincount(q_no[rows], minlength=n_q))
            renumber = np.full(n_q, -1, dtype=np.int64)
            renumber[present] = np.arange(present.shape[0])
            LOGGER.info("computed scores in %s seconds", perf_counter() - t0)
            return Ranking._assemble(cols, "fast-forward", renumber[q_no[rows]], [q_order[i] for i in present])
        # ids -> ordinals once, for the whole ranking: an unknown id fails here, before any device
        # work (the reference raises from its first batch; either way no partial result is returned)
        ordinals = self._resolve_ids(df["id"].to_numpy(), self.mode) if early_stopping is None else None
        if early_stopping is None and len(df) > 0 and (batch_size is None or batch_size >= n_q or _grouped_by_query(q_no, n_q)[0] is None):
            # the common case without the intermediate frames: CSR in, scores out
            perm, pair_off = _grouped_by_query(q_no, n_q)
            if perm is None and final_order:
                # rows of a query are adjacent and the queries already stand in their final order
                # (a Ranking the constructor sorted: checked above), so the device's per-query order
                # is the whole sort
                if batch_size is None or batch_size >= n_q:
                    dense, _, order = self._score_pairs(query_vectors, pair_off, ordinals, self.mode, sort=True)
                else:
                    # `batch_size` queries per device call (bounds the workspace, base.py:445-459)
                    dense, order = np.empty(len(df), dtype=np.float32), np.empty(len(df), dtype=np.int64)
                    for qa in range(0, n_q, batch_size):
                        qb = min(n_q, qa + batch_size)
                        pa, pb = int(pair_off[qa]), int(pair_off[qb])
                        if pb > pa:
                            part, _, part_order = self._score_pairs(query_vectors[qa:qb], pair_off[qa:qb + 1] - pa, ordinals[pa:pb],
                                                                    self.mode, sort=True)
                            dense[pa:pb], order[pa:pb] = part, part_order.astype(np.int64) + pa
                cols = {c: df[c].to_numpy()[order] for c in ("q_id", "id")}
                cols["score"] = dense[order].astype(df.dtypes["score"], copy=False)
                cols["query"] = df["query"].to_numpy()[order]
                LOGGER.info("computed scores in %s seconds", perf_counter() - t0)
                return Ranking._assemble(cols, "fast-forward", q_no, q_order)
            if perm is None:
                dense, _, _ = self._score_pairs(query_vectors, pair_off, ordinals, self.mode)
            else:
                dense, _, _ = self._score_pairs(query_vectors, pair_off, ordinals[perm], self.mode)
                unperm = np.empty_like(dense)
                unperm[perm] = dense
                dense = unperm
            result = df.copy()
            result["score"] = dense.astype(df.dtypes["score"], copy=False)
            LOGGER.info("computed scores in %s seconds", perf_counter() - t0)
            return Ranking(Ranking._trusted(result), name="fast-forward", dtype=df.dtypes["score"], copy=False, is_sorted=False)
        work = df.assign(q_no=q_no) if ordinals is None else df.assign(q_no=q_no, _ord=ordinals)

        def score(part: pd.DataFrame) -> pd.DataFrame:
            if early_stopping is None:
                return self._compute_scores(part, query_vectors)
            return self._early_stopping(part, query_vectors, early_stopping, early_stopping_alpha, early_stopping_depths)

        if batch_size is None or batch_size >= n_q:
            result = score(work)
        else:
            # bound device workspace: `batch_size` queries per launch; an empty batch is a no-op
            batch_of = work["q_no"].to_numpy() // batch_size
            result = pd.concat([score(work[batch_of == b]) for b in range((n_q + batch_size - 1) // batch_size)])
        result["score"] = result["ff_score"]
        LOGGER.info("computed scores in %s seconds", perf_counter() - t0)
        # the pairs are those of `ranking`, already known to be unique
        return Ranking(Ranking._trusted(result), name="fast-forward", dtype=df.dtypes["score"], copy=False, is_sorted=False)

    def _ordinals_of_codes(self, ranking: Ranking, codes: np.ndarray, names: np.ndarray, missing_ok: bool = False) -> np.ndarray:
        """Index ordinals of a ranking's rows through its id codes: the DISTINCT id strings are looked
        up once (and remembered on the ranking for this index and mode), the rows follow by an integer
        gather.  ``IndexError`` names the first unknown id in row order, like ``_resolve_ids``; with
        ``missing_ok`` unknown ids come back as -1 instead (early stopping looks ids up leniently)."""
        key = (id(self), getattr(getattr(self, "_ids", None), "version", None), self.mode)
        cached = getattr(ranking, "_ordinal_cache", None)
        self._last_lookup_timings = {}
        if cached is None or cached[0] != key or cached[1] is not names or cached[2] is not codes:
            t0 = perf_counter()
            by_code = self._lookup_optional(names, self.mode, ranking._id_blob(names)).astype(np.int32, copy=False)
            t1 = perf_counter()
            # a ranking's rows never change: the resolved candidate list is kept with it ("amortise /
            # cache the resolved CSR per run")
            from fast_forward_amd import native

            cached = (key, names, codes, native.take32(np.ascontiguousarray(by_code), codes), bool(by_code.size and by_code.min() < 0))
            ranking._ordinal_cache = cached
            self._last_lookup_timings = {"distinct_ids": int(len(names)), "distinct_id_lookup_s": t1 - t0, "ordinals_of_rows_s": perf_counter() - t1}
        ordinals = cached[3]
        if not missing_ok and cached[4]:
            unknown = np.flatnonzero(ordinals < 0)
            if unknown.size:  # the first one in ROW order, as _resolve_ids on the rows would name it
                raise IndexError(f"ID {names[codes[int(unknown[0])]]} not found in the index.")
        return ordinals

    def _reuse_kw(self) -> dict:
        """``reuse_outputs=True`` for backends whose ``_score_pairs`` can hand out its own result buffers (the
        caller gathers out of them before the next call)."""
        return {"reuse_outputs": True} if getattr(self, "_result_buffers", None) is not None else {}

    def _lookup_optional(self, ids: np.ndarray, mode: Mode, packed=None) -> np.ndarray:
        """Ordinals of ``ids``, -1 where unknown (``packed``: the ids once more as one joined buffer, see ``IdTable``)."""
        return self._ids.lookup_optional(ids, mode.value, packed)

    @abc.abstractmethod
    def _native_id_table(self, mode: Mode):
        """The native (``ffx_ids``) table of document or passage ids, ordinals as ``_resolve_ids``."""

    def rerank(self, ranking, alpha: float, cutoff: int | None = None):
        """Fused device path for the usual pipeline ``first.interpolate(index(first), alpha)``:
        dense scores, the alpha-interpolation (ranking.py:319) and the per-query ordering
        (ranking.py:115-117) all happen on the device; returns the interpolated ``Ranking``
        (optionally cut to the best ``cutoff`` per query).  Same scores, bit for bit."""
        if not ranking.has_queries:
            raise ValueError("Input ranking has no queries attached.")
        t = [perf_counter()]
        q_no, q_order, texts = self._numbered_queries(ranking)
        query_vectors = np.ascontiguousarray(self.encode_queries(texts))
        perm, pair_off = ranking._csr()
        t.append(perf_counter())
        if perm is None and len(q_no) >= _CODED_MIN_ROWS and _in_final_order(q_order):
            # a large run in the constructor's order: integer codes in, integer codes out, no Python string per row
            codes, names = ranking._id_coding()
            t.append(perf_counter())
            ordinals = self._ordinals_of_codes(ranking, codes, names)
            t.append(perf_counter())
            sparse = np.ascontiguousarray(ranking._scores(), dtype=np.float32)
            _, interp, order = self._score_pairs(query_vectors, pair_off, ordinals, self.mode, sparse, alpha, sort=True, want_dense=False,
                                                 **self._reuse_kw())
            t.append(perf_counter())
            from fast_forward_amd import native

            if cutoff is not None:
                rank = np.arange(len(order)) - np.repeat(pair_off[:-1], np.diff(pair_off))
                keep = rank < cutoff
                order, q_no = order[keep], q_no[keep]  # (the device order is a permutation inside every query)
            out = Ranking._assemble_coded(ranking.name, q_no, q_order, native.take32(codes, order), names,
                                          native.take32(interp, order).astype(ranking._score_dtype(), copy=False), texts)
            t.append(perf_counter())
            # where a call's time went (seconds): a first call on a ranking pays for the distinct-id look-up, later ones do not
            self.last_timings = dict(zip(("queries_numbered_and_encoded_s", "id_coding_s", "ids_to_ordinals_s", "upload_device_download_s", "result_gathers_s"),
                                         np.diff(t).tolist()), **getattr(self, "_last_lookup_timings", {}))
            return out
        df = ranking._df
        ordinals = self._resolve_ids(df["id"].to_numpy(), self.mode)
        sparse = df["score"].to_numpy(dtype=np.float32)
        if perm is not None:
            ordinals, sparse = ordinals[perm], sparse[perm]
        _, interp, order = self._score_pairs(query_vectors, pair_off, ordinals, self.mode, sparse, alpha, sort=True)
        src = order if perm is None else perm[order]
        if cutoff is not None:
            rank = np.arange(len(src)) - np.repeat(pair_off[:-1], np.diff(pair_off))
            keep = rank < cutoff
            src, order = src[keep], order[keep]
        if perm is None and _in_final_order(q_order):
            # the queries already stand in their final order (a Ranking the constructor sorted) and
            # the device ordered each one: nothing is left to sort or to check
            cols = {c: df[c].to_numpy()[src] for c in ("q_id", "id")}
            cols["score"] = interp[order].astype(df.dtypes["score"], copy=False)
            cols["query"] = df["query"].to_numpy()[src]
            return Ranking._assemble(cols, ranking.name, q_no[src], q_order)
        out = df.iloc[src].reset_index(drop=True)
        out["score"] = interp[order].astype(df.dtypes["score"])
        return Ranking(Ranking._trusted(out), name=ranking.name, dtype=df.dtypes["score"], copy=False, is_sorted=False)

    def rerank_file(self, run_in, run_out, alpha: float, queries, cutoff: int | None = None, name: str | None = None) -> dict:
        """The README pipeline of the reference, file to file, with no pandas in between:

            Ranking.from_file(run_in).attach_queries(queries).cut(cutoff)  ->  index(...)  ->
            first.interpolate(out, alpha).save(run_out)

        The run is parsed natively into the candidate CSR (ids resolved against this index while
        parsing: an unknown id raises ``IndexError``), scored, interpolated and ordered on the
        device, and written back as a TREC run.  ``queries`` maps query id -> query text.
        Returns the device stage times.  Same scores and ordering as the step-by-step API."""
        from fast_forward_amd import native

        run = native.RunFile(run_in, self._native_id_table(self.mode))
        try:
            q_ids = run.query_ids()
            missing = [q for q in q_ids if q not in queries]
            if missing:
                raise ValueError("Queries are incomplete.")
            pair_off, ordinals, sparse = run.export(cutoff or 0)
            query_vectors = np.ascontiguousarray(self.encode_queries([queries[q] for q in q_ids]))
            _, interp, order = self._score_pairs(query_vectors, pair_off, ordinals, self.mode, sparse, alpha, sort=True)
            if name is None and run.tag in _PANDAS_NA_STRINGS:
                name = "nan"  # Ranking.from_file reads such a method name as NaN and save() prints "nan"
            run.write(run_out, interp, order, tag=name)
            return dict(getattr(self, "last_stats", None) or {}, n_queries=len(q_ids), n_pairs=int(pair_off[-1]))
        finally:
            run.close()

    # ---- interchange with the reference ------------------------------------------------------------------
    def export_hdf5(self, target, chunk_size: int = 2**14, max_id_length: int | None = None) -> None:
        """Write this index as an HDF5 file the reference's ``OnDiskIndex.load`` opens
        (``index/disk.py:355-418``; layout ``:83-85, 123-165``): stored rows (vectors, or codes with
        the quantizer next to them), ids, ``num_vectors``.  ``OnDiskIndex.load`` here reads the same
        file back.  Written by :mod:`h5write` (no HDF5 library is involved)."""
        import fast_forward_amd
        from fast_forward_amd.index import h5write

        n = len(self)
        docs, psgs = self._ids.row_ids(n)
        longest = max([1] + [len(x.encode("utf-8")) for x in docs if x is not None] + [len(x.encode("utf-8")) for x in psgs if x is not None])
        if max_id_length is None:
            max_id_length = max(8, longest)
        if longest > max_id_length:
            raise RuntimeError(f"an ID is longer than the maximum ({max_id_length} characters).")
        first = next(iter(self._batch_iter(1)), None) if n else None
        quantizer = None
        if self._quantizer is not None:
            meta, attributes, data = self._quantizer.serialize()
            # class names are shared with the reference; its classes live in fast_forward.quantizer.nanopq
            meta = dict(meta, __module__="fast_forward.quantizer.nanopq" if meta["__name__"] in ("NanoPQ", "NanoOPQ") else meta["__module__"])
            quantizer = (meta, {k: v for k, v in attributes.items() if v is not None}, data)
            dim, dtype = self._quantizer.dims[1], np.dtype(self._quantizer.dtype)
        elif first is not None:
            dim, dtype = first[0].shape[1], first[0].dtype
        else:
            raise ValueError("an empty index has no dimensionality to write")
        h5write.write_index(target, dim=int(dim), dtype=dtype, chunk_size=int(chunk_size), max_id_length=int(max_id_length),
                            num_vectors=n, vector_blocks=(rows for rows, _, _ in self._batch_iter(max(chunk_size, 2**14))),
                            doc_ids=docs, psg_ids=psgs, ff_version=getattr(fast_forward_amd, "__version__", "0"), quantizer=quantizer)

    # ---- iteration ----------------------------------------------------------------------------------------
    def batch_iter(self, batch_size: int) -> Iterator[tuple[np.ndarray, IDSequence, IDSequence]]:
        """Vectors (decoded, if a quantizer is attached), document ids and passage ids in
        insertion order, ``batch_size`` at a time."""
        if self._quantizer is None:
            yield from self._batch_iter(batch_size)
        else:
            for codes, doc_ids, psg_ids in self._batch_iter(batch_size):
                yield self._quantizer.decode(codes), doc_ids, psg_ids

    def __iter__(self) -> Iterator[tuple[np.ndarray, "str | None", "str | None"]]:
        for vectors, doc_ids, psg_ids in self.batch_iter(2**9):
            yield from zip(vectors, doc_ids, psg_ids)
