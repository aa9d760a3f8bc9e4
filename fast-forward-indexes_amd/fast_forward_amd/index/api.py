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


def _drop_kept_scores(index_ref, token: int) -> None:
    """Finaliser of a ranking that came out of ``Index.__call__``: if the index still keeps the dense scores of that very call
    (nobody has displaced them), release them -- nothing else could ask for them (``Ranking.interpolate`` needs that ranking)."""
    index = index_ref()
    try:
        if index is not None and getattr(index, "_store", None) is not None and index._kept_token() == token:
            from fast_forward_amd import native

            native.drop_kept(index._store)
    except Exception:  # noqa: BLE001  (interpreter shutdown, a closed index: nothing left to release)
        pass

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


class _Run:
    """A ranking as an index scores it: integers only.  Queries numbered in order of first appearance (base.py:418-421)
    with their vectors, rows GROUPED by query number (``perm``: the rows of ``ranking`` in that layout, None when they
    already stand so; ``pair_off``: where every query's candidates start), ids as codes into a table of distinct
    strings.  Whatever the ranking holds inside (a coded form or a pandas frame), this is the one form the scoring
    routes see, and ``result`` is the one way back: a coded ``Ranking`` (its frame of strings is built if and when
    somebody asks for it)."""

    def __init__(self, index: "Index", ranking: Ranking, encode: bool = True):
        self.ranking = ranking
        self.q_no, self.q_order, self.texts = index._numbered_queries(ranking)
        self.n_q = len(self.q_order)
        if not encode:  # (a route that reads no query vector: Index._interpolate_kept)
            self.query_vectors = None
        else:
            self.query_vectors = np.ascontiguousarray(index.encode_queries(self.texts)) if self.n_q else np.zeros((0, index.dim or 0), dtype=np.float32)
        self.perm, self.pair_off = ranking._csr() if len(self.q_no) else (None, np.zeros(self.n_q + 1, dtype=np.int64))
        self.codes, self.names = ranking._id_coding()

    def grouped(self, per_row: np.ndarray) -> np.ndarray:
        """A per-row array of the ranking in the grouped layout."""
        if self.perm is None:
            return per_row
        from fast_forward_amd import native

        return native.take32(np.ascontiguousarray(per_row), self.perm.astype(np.int32)) if per_row.dtype.itemsize == 4 else per_row[self.perm]

    def batches(self, batch_size: "int | None"):
        """(first query, one past the last) of every device call: all queries at once, or ``batch_size`` at a time
        (base.py:445-459; an empty batch is a no-op here, the reference raises on one)."""
        step = self.n_q if (batch_size is None or batch_size >= self.n_q) else max(1, int(batch_size))
        return [(qa, min(self.n_q, qa + step)) for qa in range(0, self.n_q, max(1, step))]

    def result(self, name, kept: "np.ndarray | None", scores: np.ndarray, off: np.ndarray, q_range: "tuple[int, int] | None" = None,
               codes: "np.ndarray | None" = None) -> Ranking:
        """The ``Ranking`` of the rows ``kept`` (positions in the grouped layout; query after query in query-number
        order, ``off[q]`` .. ``off[q+1]`` those of query q of ``q_range``, inside a query in their final order) with
        ``scores``.  Queries without a kept row disappear; the queries are put in descending id order if they do not
        stand so already (ranking.py:115-117).  ``codes``: the id codes of the kept rows when the device has permuted
        them along already (``kept`` may then be None: nothing is gathered here)."""
        from fast_forward_amd import native

        q_lo, q_hi = q_range if q_range is not None else (0, self.n_q)
        counts = np.diff(off)
        present = np.flatnonzero(counts)
        q_order = [self.q_order[q_lo + i] for i in present]
        src = None if kept is None else (kept if self.perm is None else self.perm[kept])  # rows of the ranking
        if not _in_final_order(q_order):  # lay the queries' blocks out in descending id order
            by_name = sorted(range(len(present)), key=lambda i: q_order[i], reverse=True)
            blocks = present[by_name]
            pick = np.concatenate([np.arange(off[q], off[q + 1]) for q in blocks]) if len(blocks) else np.zeros(0, dtype=np.int64)
            src, scores, present, q_order = (None if src is None else src[pick]), scores[pick], blocks, [q_order[i] for i in by_name]
            codes = None if codes is None else codes[pick]
            counts_out, moved = counts[blocks], True
        else:
            counts_out, moved = counts[present], False
        if q_range is None and self.perm is None and not moved and len(present) == self.n_q and len(scores) == len(self.q_no) \
                and np.array_equal(off, self.pair_off):
            q_no = self.q_no  # every row kept, the queries' blocks where they were: the ranking's own numbering (35 M int64 not written again)
        else:
            q_no = native.run_numbers(np.concatenate([[0], np.cumsum(counts_out)]))
        texts = None if self.texts is None else [self.texts[q_lo + i] for i in present]
        if codes is None:
            codes = native.take32(self.codes, src) if src.dtype == np.int32 else self.codes[src]
        out = Ranking._assemble_coded(name, q_no, q_order, codes, self.names, scores.astype(self.ranking._score_dtype(), copy=False), texts,
                                      self.ranking._id_blob(self.names))
        return out, src


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
        whole = getattr(self.query_encoder, "encode_all", None)
        if whole is not None and len(queries):  # (an encoder that is a table, not a model: one gather, see ArrayEncoder)
            return whole(queries)
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
        every row in the current mode (the contract of base.py:279-314, computed on the device).  The frame form of
        ``_score_pairs`` for code written against the reference's internals; ``__call__`` itself works on integers."""
        out = data.copy()
        if len(out) == 0:
            out["ff_score"] = np.zeros(0, dtype=np.float32)
            return out
        q_no = out["q_no"].to_numpy(dtype=np.int64)
        ordinals = self._resolve_ids(out["id"].to_numpy(), self.mode)
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
        out = df.copy()
        out["ff_score"] = self._early_stopping_scores(df["id"].to_numpy(), df["q_no"].to_numpy(dtype=np.int64), df["score"].to_numpy(),
                                                      query_vectors, cutoff, alpha, depths)
        return out

    def _early_stopping_scores(self, ids: np.ndarray, q_no: np.ndarray, sparse: np.ndarray, query_vectors: np.ndarray,
                               cutoff: int, alpha: float, depths: Iterable[int]) -> np.ndarray:
        """Array form of early stopping (base.py:316-387): the dense scores in row order, NaN for the candidates that
        were never scored.  Rows of a query need not be adjacent; inside a query they stand in first-stage rank order."""
        n_q = query_vectors.shape[0]
        perm, pair_off = _grouped_by_query(np.asarray(q_no), n_q)
        ids_g = ids if perm is None else ids[perm]
        rows, kept = self._early_stopping_kept(lambda p: ids_g[p], sparse if perm is None else sparse[perm], pair_off, query_vectors, cutoff, alpha,
                                               depths, self._lookup_optional(ids_g, self.mode).astype(np.int32))
        ff = np.full(len(q_no), np.nan, dtype=np.float32)
        ff[rows if perm is None else perm[rows]] = kept
        return ff

    def _early_stopping_kept(self, id_at, sparse: np.ndarray, pair_off: np.ndarray, query_vectors: np.ndarray, cutoff: int, alpha: float,
                             depths: Iterable[int], ordinals: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        """Score each query only as deep as needed for its top-``cutoff`` (base.py:316-387): (the candidates that got
        scored, ascending; their dense scores).  Candidates stand grouped by query (``pair_off``), in first-stage rank
        order inside a query; ``ordinals`` may hold -1 for an id the index does not know -- as in the reference (which
        looks ids up chunk by chunk, base.py:362-370) that is an ``IndexError`` only if the candidate is reached
        (``id_at(position)`` names it).

        Depth intervals ``[a, b)`` over the sorted depths ``>= cutoff``; after the first interval a query continues only
        while its ``cutoff``-th best interpolated score is below the best it could still reach,
        ``alpha * s_last + (1 - alpha) * max_dense`` (``s_last``: first-stage score of its last scored candidate),
        evaluated under ``early_stopping_promotion``; an interval without candidates ends the loop.  This is the NumPy
        statement; ``InMemoryIndex`` keeps all of it on the device."""
        n, n_q = int(pair_off[-1]), query_vectors.shape[0]
        q_of = np.repeat(np.arange(n_q, dtype=np.int64), np.diff(pair_off))
        rank_in_q = np.arange(n, dtype=np.int64) - pair_off[q_of]
        bound_f32 = self._es_bound_f32()
        ff = np.full(n, np.nan, dtype=np.float32)
        interp = np.full(n, np.nan, dtype=np.result_type(sparse.dtype, np.float32))
        active = np.ones(n_q, dtype=bool)
        a = 0
        for b in sorted(depths):
            if b < cutoff:
                continue
            if a > 0:
                # continue-criterion for all queries at once, over the scored prefix [0, min(len, a)) of every query
                ends = np.minimum(pair_off[1:], pair_off[:-1] + a)
                present = np.flatnonzero(active & (pair_off[1:] > pair_off[:-1]))
                rows = np.flatnonzero(active[q_of] & (rank_in_q < a))  # (every row of a still-active query below depth a has been scored)
                qs = q_of[rows]
                starts = np.flatnonzero(np.r_[True, qs[1:] != qs[:-1]])
                stops = np.r_[starts[1:], rows.size]
                by_score = rows[np.lexsort((-interp[rows].astype(np.float64), qs))]
                kth = interp[by_score[np.minimum(starts + cutoff, stops) - 1]]  # cutoff-th best so far (the worst, if fewer)
                last_sparse = sparse[ends[present] - 1]
                best_dense = np.maximum.reduceat(ff[rows], starts)
                if bound_f32:  # (arrays times a Python float keep their dtype under either NumPy)
                    bound = alpha * last_sparse + (1 - alpha) * best_dense
                else:
                    bound = np.float64(alpha) * last_sparse.astype(np.float64) + np.float64(1 - alpha) * best_dense.astype(np.float64)
                keep = np.zeros(n_q, dtype=bool)
                keep[present] = kth < bound
                active &= keep
            LOGGER.info("depth %s: %s queries left", b, int(active.sum()))
            take = np.flatnonzero(active[q_of] & (rank_in_q >= a) & (rank_in_q < b))
            if take.size == 0:
                break
            unknown = np.flatnonzero(ordinals[take] < 0)
            if unknown.size:
                raise IndexError(f"ID {id_at(int(take[unknown[0]]))} not found in the index.")
            part_off = np.zeros(n_q + 1, dtype=np.int64)
            np.cumsum(np.bincount(q_of[take], minlength=n_q), out=part_off[1:])
            chunk_ff, _, _ = self._score_pairs(query_vectors, part_off, np.ascontiguousarray(ordinals[take], dtype=np.int32), self.mode)
            ff[take] = chunk_ff
            interp[take] = alpha * sparse[take] + (1 - alpha) * chunk_ff
            a = b
        rows = np.flatnonzero(~np.isnan(ff))
        return rows, ff[rows]

    def __call__(self, ranking: Ranking, early_stopping: int | None = None, early_stopping_alpha: float | None = None,
                 early_stopping_depths: Iterable[int] | None = None, batch_size: int | None = None) -> Ranking:
        """Dense scores for every (query, candidate) of ``ranking`` (base.py:389-469).

        ``ValueError``: ranking without queries; early stopping without alpha/depths.
        ``IndexError``: a candidate id that is not in the index (raised before any device work; with early stopping, as
        in the reference, only for a candidate that gets scored).

        One route for every ranking: ids travel as integer codes (``_Run``), every batch of queries is one device call
        that scores and orders (``_score_pairs``, or ``_early_stopping_kept``), and the result is assembled in coded
        form -- no Python string per row on the way in or out; its frame is built on first use."""
        if not ranking.has_queries:
            raise ValueError("Input ranking has no queries attached.")
        if early_stopping is not None and (early_stopping_alpha is None or early_stopping_depths is None):
            raise ValueError("Early stopping requires alpha and depths.")
        from fast_forward_amd import native

        t0 = perf_counter()
        run = _Run(self, ranking)
        off = run.pair_off
        kept_parts, score_parts = [], []
        if early_stopping is None:
            whole = len(run.batches(batch_size)) == 1
            n_rows = len(run.q_no)
            ranked = whole and hasattr(self, "_score_ranked") and n_rows >= self._RANKED_FROM
            warm = self._warm_up(run.n_q, n_rows, False, [(n_rows, np.float32), (n_rows, np.int32), (n_rows, np.int32)]) if ranked else None
            ordinals = run.grouped(self._ordinals_of_codes(ranking, run.codes, run.names))
            if warm is not None:
                warm.join()
            if ranked:
                # the device lays the run out in its final order, the ranking's id codes permuted along: nothing is gathered here
                # ... and the dense scores stay in HBM beside the codes (8 bytes per candidate): the `ranking.interpolate(result,
                # alpha)` that usually follows is then the interpolation and the ordering alone (_interpolate_kept)
                keeps = hasattr(self, "_rank_kept")
                scores, codes, kept = self._score_ranked(run.query_vectors, off, ordinals, self.mode, None, None, run.grouped(run.codes), want_order=True,
                                                         **({"keep": True} if keeps else {}))
                out, _ = run.result("fast-forward", kept, scores, off, codes=codes)
                LOGGER.info("computed scores in %s seconds", perf_counter() - t0)
                src = kept if run.perm is None else run.perm[kept]
                out._source_rows = (weakref.ref(ranking), src)
                token = self._kept_token() if keeps else 0
                out._origin = (weakref.ref(self), self.mode, getattr(getattr(self, "_ids", None), "version", None), token)
                if token:  # the kept scores (8 bytes per candidate of HBM) go when the only ranking that can use them has gone
                    weakref.finalize(out, _drop_kept_scores, weakref.ref(self), token)
                return out
            for qa, qb in run.batches(batch_size):
                pa, pb = int(off[qa]), int(off[qb])
                dense, _, order = self._score_pairs(run.query_vectors[qa:qb], off[qa:qb + 1] - pa, ordinals[pa:pb], self.mode, sort=True,
                                                    **(self._reuse_kw() if whole else {}))
                kept_parts.append(order if pa == 0 else order + np.int32(pa))
                score_parts.append(native.take32(dense, order))
            kept_off = off
        else:
            lenient = run.grouped(self._ordinals_of_codes(ranking, run.codes, run.names, missing_ok=True))
            sparse, codes_g = run.grouped(ranking._scores()), run.grouped(run.codes)
            for qa, qb in run.batches(batch_size):
                pa, pb = int(off[qa]), int(off[qb])
                rows, kept = self._early_stopping_kept(lambda p, pa=pa: run.names[codes_g[pa + p]], sparse[pa:pb], off[qa:qb + 1] - pa,
                                                       run.query_vectors[qa:qb], early_stopping, early_stopping_alpha, early_stopping_depths, lenient[pa:pb])
                kept_parts.append(rows.astype(np.int32) + np.int32(pa))
                score_parts.append(kept)
            rows = np.concatenate(kept_parts) if kept_parts else np.zeros(0, dtype=np.int32)
            scores = np.concatenate(score_parts).astype(np.float32, copy=False) if score_parts else np.zeros(0, dtype=np.float32)
            kept_off = np.searchsorted(rows, off).astype(np.int64)  # rows ascend: still grouped by query
            order = native.host_order(scores, kept_off)  # score descending inside every query, exact ties by first-stage rank
            kept_parts, score_parts = [native.take32(rows, order)], [native.take32(scores, order)]
        kept = kept_parts[0] if len(kept_parts) == 1 else (np.concatenate(kept_parts) if kept_parts else np.zeros(0, dtype=np.int32))
        scores = score_parts[0] if len(score_parts) == 1 else (np.concatenate(score_parts) if score_parts else np.zeros(0, dtype=np.float32))
        out, src = run.result("fast-forward", kept, scores, kept_off)
        LOGGER.info("computed scores in %s seconds", perf_counter() - t0)
        if early_stopping is None:
            # row i of the result is row src[i] of `ranking`: ranking.interpolate(result, alpha) needs no join.  The result owns
            # that array (a view into the index's re-used result block is copied: round 4 let the result borrow the block
            # until the index needed it again -- one more piece of state between two objects for 4 bytes per row saved on a
            # route that large runs do not take: from 256 K candidates on the device applies the ordering, above)
            out._source_rows = (weakref.ref(ranking), src if src.base is None else np.array(src, dtype=np.int32))
        return out

    _RANKED_FROM = 1 << 18  # candidates from which the device applies the ordering itself (ffx_score_ranked)

    def _warm_up(self, n_queries: int, n_rows: int, has_sparse: bool, result_blocks: list):
        """A helper thread that gets ready what a large device call will need while this thread resolves the ranking's ids
        (host work): the page-locked blocks of the result pool the result will live in; a resident index adds its own
        device-side preparation.  ``join`` it before the device call."""
        from fast_forward_amd import native

        return native.reserve_pooled(result_blocks)

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

            # (ordinary memory, although the list goes to the device on every call: page-locking 140 MB here would add ~35 ms
            # to the FIRST call on a ranking -- measured: 0.21 -> 0.31 s with the result blocks competing for the pool -- to save
            # ~1.5 ms on each later one; the ranking's own scores and codes are page-locked by its constructor)
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
        from fast_forward_amd import native

        t = [perf_counter()]
        n_rows = len(ranking._q_no)
        ranked = hasattr(self, "_score_ranked") and n_rows >= self._RANKED_FROM
        # the blocks the result will live in are page-locked (once) while the queries are encoded and the ids looked up
        warm = self._warm_up(len(ranking._q_order), n_rows, True, [(n_rows, np.float32), (n_rows, np.int32)]) if ranked else None
        run = _Run(self, ranking)
        t.append(perf_counter())
        ordinals = run.grouped(self._ordinals_of_codes(ranking, run.codes, run.names))
        if warm is not None:
            warm.join()
        t.append(perf_counter())
        sparse = np.ascontiguousarray(run.grouped(ranking._scores()), dtype=np.float32)
        off = run.pair_off
        if ranked:
            # the device applies the ordering: interpolated scores and the ranking's id codes come back in final order, in
            # page-locked blocks of the result pool that the result keeps -- no gather, no copy, no fresh pages on the host
            carry = run.grouped(run.codes)
            t.append(perf_counter())
            scores, codes, _ = self._score_ranked(run.query_vectors, off, ordinals, self.mode, sparse, alpha, carry)
            t.append(perf_counter())
            if cutoff is not None:  # (a query's block stands in final order: its first `cutoff` rows are its best)
                keep = np.arange(len(scores)) - np.repeat(off[:-1], np.diff(off)) < cutoff
                scores, codes = scores[keep], codes[keep]
                off = np.concatenate([[0], np.cumsum(np.minimum(np.diff(off), cutoff))]).astype(np.int64)
            out, _ = run.result(ranking.name, None, scores, off, codes=codes)
            t.append(perf_counter())
            names = ("queries_numbered_and_encoded_s", "ids_to_ordinals_s", "first_stage_scores_and_codes_grouped_s", "upload_device_download_s", "result_assembled_s")
        else:
            _, interp, order = self._score_pairs(run.query_vectors, off, ordinals, self.mode, sparse, alpha, sort=True, want_dense=False, **self._reuse_kw())
            t.append(perf_counter())
            if cutoff is not None:  # (the device order is a permutation inside every query)
                keep = np.arange(len(order)) - np.repeat(off[:-1], np.diff(off)) < cutoff
                order = order[keep]
                off = np.concatenate([[0], np.cumsum(np.minimum(np.diff(off), cutoff))]).astype(np.int64)
            out, _ = run.result(ranking.name, order, native.take32(interp, order), off)
            t.append(perf_counter())
            names = ("queries_numbered_and_encoded_s", "ids_to_ordinals_s", "upload_device_download_s", "result_gathers_s")
        # where a call's time went (seconds): a first call on a ranking pays for the distinct-id look-up, later ones do not
        self.last_timings = dict(zip(names, np.diff(t).tolist()), total_s=t[-1] - t[0], **getattr(self, "_last_lookup_timings", {}))
        return out

    def _interpolate_kept(self, ranking, token: int, alpha: float):
        """``ranking.interpolate(self(ranking), alpha)`` when the device still keeps the dense scores of that
        ``self(ranking)`` under ``token`` (and the ranking's id codes beside them): the first-stage scores go up, the
        interpolation (ranking.py:319) and the per-query ordering (ranking.py:115-117) run on the device, interpolated
        scores and codes come back in final order.  No query is encoded, no row of the index is read; the bits are
        ``rerank``'s.  The ranking's rows must stand grouped by query (they do: the caller checked)."""
        t = [perf_counter()]
        run = _Run(self, ranking, encode=False)
        sparse = np.ascontiguousarray(run.grouped(ranking._scores()), dtype=np.float32)
        t.append(perf_counter())
        scores, codes = self._rank_kept(token, run.pair_off, sparse, alpha)
        t.append(perf_counter())
        out, _ = run.result(ranking.name, None, scores, run.pair_off, codes=codes)
        t.append(perf_counter())
        self.last_timings = dict(zip(("queries_numbered_s", "upload_device_download_s", "result_assembled_s"), np.diff(t).tolist()), total_s=t[-1] - t[0])
        return out

    def rerank_file(self, run_in, run_out, alpha: float, queries, cutoff: int | None = None, name: str | None = None) -> dict:
        """The README pipeline of the reference, file to file, with no pandas in between:

            Ranking.from_file(run_in).attach_queries(queries).cut(cutoff)  ->  index(...)  ->
            first.interpolate(out, alpha).save(run_out)

        The run is parsed natively into the candidate CSR (ids resolved against this index while
        parsing: an unknown id raises ``IndexError``), scored, interpolated and ordered on the
        device, and written back as a TREC run.  ``queries`` maps query id -> query text.
        Returns the device stage times.  Same scores and ordering as the step-by-step API."""
        from fast_forward_amd import native

        t = [perf_counter()]
        run = native.RunFile(run_in, self._native_id_table(self.mode))
        t.append(perf_counter())
        parse_stages = native.stage_report()
        try:
            q_ids = run.query_ids()
            missing = [q for q in q_ids if q not in queries]
            if missing:
                raise ValueError("Queries are incomplete.")
            pair_off, ordinals, sparse = run.export(cutoff or 0)
            t.append(perf_counter())
            query_vectors = np.ascontiguousarray(self.encode_queries([queries[q] for q in q_ids]))
            t.append(perf_counter())
            _, interp, order = self._score_pairs(query_vectors, pair_off, ordinals, self.mode, sparse, alpha, sort=True)
            t.append(perf_counter())
            if name is None and run.tag in _PANDAS_NA_STRINGS:
                name = "nan"  # Ranking.from_file reads such a method name as NaN and save() prints "nan"
            run.write(run_out, interp, order, tag=name)
            t.append(perf_counter())
            # where the call's time went (seconds), the parser's and the writer's own stages beside it
            self.last_file_timings = dict(zip(("parse_and_resolve_s", "export_s", "encode_queries_s", "upload_device_download_s", "write_s"), np.diff(t).tolist()),
                                          parse_stages=parse_stages, write_stages=native.stage_report())
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
