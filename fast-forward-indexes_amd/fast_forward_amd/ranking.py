"""``Ranking``: a TREC run (query id, document/passage id, score) with the reference's API.

Mirrors ``fast_forward.Ranking`` (reference ``src/fast_forward/ranking.py:64-409``): same
constructor arguments, methods, exceptions and score arithmetic, so code written against the
reference runs unchanged.  It stays a host-side container on a pandas frame (``_df`` with
columns ``q_id``, ``id``, ``score`` and optionally ``query``); the device work of this
package happens in ``fast_forward_amd.index``.

Arithmetic notes (pinned by tests/golden/interpolate.npz, generated from the reference):
  * scores are ``dtype`` (float32 by default); ``alpha * a + (1 - alpha) * b`` rounds both
    products and the sum to float32, with ``1 - alpha`` formed in double (ranking.py:319)
  * a pair missing from either side of ``interpolate`` / ``+`` counts as score 0 (ranking.py:318)
  * ordering is query id descending, then score descending; exact ties keep their input
    order here (the reference leaves tie order unspecified, ranking.py:115-117)
"""

from __future__ import annotations

import logging
from collections.abc import Iterator, Mapping
from pathlib import Path
from time import perf_counter

import numpy as np
import pandas as pd

from fast_forward_amd.names import Names

LOGGER = logging.getLogger(__name__)

Run = Mapping[str, Mapping[str, float]]
_KEY = ["q_id", "id"]
_CODED_FROM_FILE_ROWS = 1 << 16  # run files with at least this many rows are read into the coded form
BUILD_TIMINGS: dict = {}  # where the time of the last large Ranking(...) / Ranking.from_columns went (seconds per stage)


def _with_queries(df: pd.DataFrame, queries: Mapping[str, str]) -> pd.DataFrame:
    """Add the ``query`` column; every query id of the frame must be covered (ranking.py:16-28)."""
    missing = set(df["q_id"].unique()) - set(queries)
    if missing:
        raise ValueError("Queries are incomplete.")
    out = df.drop(columns=["query"], errors="ignore")
    out["query"] = out["q_id"].map(dict(queries))
    return out


def _scaled_to_unit_range(df: pd.DataFrame) -> pd.DataFrame:
    """Global (not per-query) min-max scaling of ``score``; constant scores become 0
    (ranking.py:45-61)."""
    out = df.copy()
    lo, hi = out["score"].min(), out["score"].max()
    if lo == hi:
        LOGGER.warning("all scores are equal, setting scores to 0")
        out["score"] = 0
    else:
        out["score"] = (out["score"] - lo) / (hi - lo)
    return out


def _ranks_within_query(df: pd.DataFrame) -> pd.Series:
    """1-based position of every row inside its query, in frame order (ranking.py:31-42)."""
    return df.groupby("q_id").cumcount() + 1


def _is_plain_text(token: str) -> bool:
    """A run name pandas reads as the string it is (not a number, boolean or NA word)."""
    if token in _PANDAS_NA_WORDS or token.lower() in ("true", "false"):
        return False
    try:
        float(token)
        return False
    except ValueError:
        return True


_PANDAS_NA_WORDS = frozenset(("", "#N/A", "#N/A N/A", "#NA", "-1.#IND", "-1.#QNAN", "-NaN", "-nan", "1.#IND", "1.#QNAN", "<NA>",
                              "N/A", "NA", "NULL", "NaN", "None", "n/a", "nan", "null"))


def _sorted_frame(df: pd.DataFrame) -> pd.DataFrame:
    """q_id descending, score descending, ties in input order (stable)."""
    q_codes = pd.factorize(df["q_id"], sort=True)[0]
    order = np.lexsort((-df["score"].to_numpy(dtype=np.float64), -q_codes))
    return df.iloc[order]


def _native_or_none():
    """The ctypes layer if libffx can be loaded (host-side helpers need no GPU), else None."""
    try:
        from fast_forward_amd import native

        native.lib()
        return native
    except (ImportError, OSError):
        return None


def _order_by_query_then_score(q_rank: np.ndarray, score: np.ndarray) -> np.ndarray:
    """Stable order by (query rank ascending, score descending): ``np.lexsort((-score, q_rank))``.  For float32
    scores every query is ordered by libffx's host threads (``ffx_host_order``) and the queries' blocks are then
    put in rank order: rows that stand grouped by query (any run file, any frame made from one -- in whatever
    order the queries come) need no sort over all rows at all; only rows of one query scattered over the input
    are first gathered by a stable integer sort.  34.9 M rows: ~0.5 s instead of ~8 s."""
    if score.dtype == np.float32 and score.shape[0] >= (1 << 16):
        try:
            from fast_forward_amd import native

            native.lib()
        except (ImportError, OSError):
            native = None
        if native is not None:
            n = q_rank.shape[0]
            starts = np.flatnonzero(q_rank[1:] != q_rank[:-1]) + 1
            starts = np.concatenate([[0], starts])
            run_rank = q_rank[starts]
            if np.unique(run_rank).shape[0] == run_rank.shape[0]:
                # one contiguous block per query: order inside every block, then lay the blocks out by rank
                off = np.concatenate([starts, [n]]).astype(np.int64)
                if (run_rank[1:] > run_rank[:-1]).all():
                    return native.host_order(score, off)
                by_rank = np.argsort(run_rank, kind="stable")
                dst = np.empty(len(by_rank), dtype=np.int64)
                dst[by_rank] = np.concatenate([[0], np.cumsum(np.diff(off)[by_rank])[:-1]])  # where block q lands
                return native.host_order(score, off, dst_off=dst)
            n_q = int(q_rank.max()) + 1
            off = np.zeros(n_q + 1, dtype=np.int64)
            np.cumsum(np.bincount(q_rank, minlength=n_q), out=off[1:])
            grouped = np.argsort(q_rank, kind="stable")
            return grouped[native.host_order(score[grouped], off)]
    return np.lexsort((-score.astype(np.float64), q_rank))


def _coded_text_column(native, col) -> "tuple[np.ndarray, object] | None":
    """(int32 code of every row, the distinct strings by code) of a column of strings in any columnar form -- a
    ``(codes, names)`` pair or pandas ``Categorical`` (pre-coded), a pyarrow string array or a pandas Series backed by one,
    a NumPy ``S`` / ``U`` array, an object array of ``str`` -- or None when the column is something else or holds a missing
    value.  No Python object per row is made for any of them (the object array HAS one per row already: its cells are
    numbered through their addresses, ``native.code_object_column``)."""
    if isinstance(col, tuple) and len(col) == 2:
        codes, names = col
        codes = np.asarray(codes)
        if codes.ndim != 1 or codes.dtype.kind not in "iu":
            return None
        if isinstance(names, Names):
            table, of_name = names, None
        else:  # the names in any column form; numbered like a column (a table may name one string twice)
            got = _coded_text_column(native, names)
            if got is None:
                return None
            of_name, table = got
        n_given = len(of_name) if of_name is not None else len(table)
        if len(codes) and (int(codes.min()) < 0 or int(codes.max()) >= n_given):
            return None  # (-1: pandas' code for a missing value)
        codes = np.ascontiguousarray(codes, dtype=np.int32)
        if of_name is not None and not (len(of_name) == len(table) and np.array_equal(of_name, np.arange(len(table), dtype=np.int32))):
            codes = native.take32(np.ascontiguousarray(of_name, dtype=np.int32), codes)
        return codes, table
    if isinstance(col, pd.Series):
        if isinstance(col.dtype, pd.CategoricalDtype):
            col = col.array
        elif col.dtype == object:
            col = col.to_numpy()
        elif hasattr(col.array, "__arrow_array__"):
            col = col.array
        else:
            return None
    if isinstance(col, pd.Categorical):
        return _coded_text_column(native, (col.codes, col.categories.to_numpy(dtype=object)))
    if isinstance(col, (list, tuple)):
        arr = np.empty(len(col), dtype=object)
        arr[:] = col
        col = arr
    if isinstance(col, np.ndarray):
        if col.dtype == object:
            if col.ndim != 1:
                return None
            got = native.code_object_column(np.ascontiguousarray(col), names_as_objects=False)
            if got is None:
                return None
            codes, names, packed = got  # (names is None where the joined buffer stands for the table: strings on demand)
            return codes, (names if packed is None else Names(packed[0], packed[1], objects=names))
        return native.code_fixed_strings(col) if col.dtype.kind in "SU" else None
    try:
        import pyarrow as pa
    except ImportError:
        return None
    if hasattr(col, "__arrow_array__") and not isinstance(col, (pa.Array, pa.ChunkedArray)):
        try:
            col = pa.array(col) if not hasattr(col, "_pa_array") else col._pa_array
        except (pa.ArrowInvalid, pa.ArrowTypeError, TypeError, ValueError):
            return None
    if isinstance(col, (pa.Array, pa.ChunkedArray)):
        return native.code_arrow_strings(col)
    return None


class Ranking:
    """Rankings of documents/passages w.r.t. queries (drop-in for ``fast_forward.Ranking``).

    Two representations of the same rows.  The *frame* (``_df``: pandas, string columns) is what the
    reference has and what every method here works on.  The *coded* form keeps what the device
    produced or needs -- per row a query number, an int32 id code and a score, plus the tables that
    turn numbers back into strings -- so that ``Index.rerank`` / ``Index.__call__`` on a large run
    never touch a Python string per row; a coded ranking grows its frame the first time somebody
    asks for it."""

    _frame: "pd.DataFrame | None" = None
    _coded: "dict | None" = None      # q_no, q_order, id_codes, id_names, score, q_texts (None: no queries)
    _id_cache: "tuple | None" = None  # (id code of every frame row, id strings by code): valid for the current frame

    @property
    def _df(self) -> pd.DataFrame:
        if self._frame is None:
            c = self._coded
            q_names = np.empty(len(c["q_order"]), dtype=object)
            q_names[:] = c["q_order"]
            cols = {"q_id": q_names[c["q_no"]] if len(q_names) else np.empty(0, dtype=object),
                    "id": c["id_names"][c["id_codes"]] if len(c["id_names"]) else np.empty(0, dtype=object), "score": c["score"]}
            if c["q_texts"] is not None:
                texts = np.empty(len(c["q_texts"]), dtype=object)
                texts[:] = c["q_texts"]
                cols["query"] = texts[c["q_no"]] if len(texts) else np.empty(0, dtype=object)
            self._frame = pd.DataFrame(cols, copy=False)
        return self._frame

    @_df.setter
    def _df(self, frame: pd.DataFrame) -> None:
        self._frame = frame
        self._coded = None
        self._id_cache = None

    @classmethod
    def _assemble_coded(cls, name, q_no: np.ndarray, q_order: list, id_codes: np.ndarray, id_names: np.ndarray, score: np.ndarray,
                        q_texts: "list | None", id_blob: "tuple[bytes, int] | None" = None) -> "Ranking":
        """Ranking from device results in coded form (rows in final order, pairs unique, nothing missing);
        the frame is built on first use.  ``id_blob``: ``id_names`` once more as one joined buffer (``_id_blob``)."""
        out = cls.__new__(cls)
        out.name = name
        out._coded = {"q_no": q_no, "q_order": list(q_order), "id_codes": id_codes, "id_names": id_names, "score": score,
                      "q_texts": None if q_texts is None else list(q_texts), "id_blob": id_blob}
        out._q_no, out._q_order = q_no, list(q_order)
        out._q_ids = set(out._q_order)
        return out

    def _id_coding(self) -> tuple[np.ndarray, np.ndarray]:
        """(int32 code of every row's id, id strings by code)."""
        if self._coded is not None:
            return self._coded["id_codes"], self._coded["id_names"]
        if self._id_cache is None:
            codes, names = pd.factorize(self._frame["id"].to_numpy())
            self._id_cache = (codes.astype(np.int32), np.asarray(names, dtype=object))
        return self._id_cache

    def _id_blob(self, names: np.ndarray) -> "tuple[bytes, int] | None":
        """The id strings of ``_id_coding`` once more as (one ``'\n'``-joined UTF-8 buffer, their number), if this
        ranking was built from such a buffer (the native constructors number ids out of one): an index resolves
        the distinct ids from it without touching a Python string."""
        if isinstance(names, Names):  # (a table that IS one buffer: Ranking.from_file, the native constructors)
            return names.packed
        c = self._coded
        if c is not None and c.get("id_blob") is not None and names is c["id_names"]:
            return c["id_blob"]
        return None

    def _scores(self) -> np.ndarray:
        return self._coded["score"] if self._coded is not None else self._frame["score"].to_numpy()

    def _score_dtype(self) -> np.dtype:
        return self._scores().dtype

    def _csr(self) -> tuple["np.ndarray | None", np.ndarray]:
        """(permutation that makes every query's rows adjacent, or None when they already are; offsets of
        every query number).  Rows never change after construction, so this is computed once."""
        cached = getattr(self, "_csr_cache", None)
        if cached is None or cached[0] is not self._q_no:
            q_no = np.asarray(self._q_no)
            pair_off = np.zeros(len(self._q_order) + 1, dtype=np.int64)
            np.cumsum(np.bincount(q_no, minlength=len(self._q_order)), out=pair_off[1:])
            perm = None if (q_no.shape[0] < 2 or (q_no[1:] >= q_no[:-1]).all()) else np.argsort(q_no, kind="stable")
            cached = (self._q_no, perm, pair_off)
            self._csr_cache = cached
        return cached[1], cached[2]

    def _query_texts(self) -> "list | None":
        """Query text per query number, or None when no queries are attached."""
        if self._coded is not None:
            return self._coded["q_texts"]
        if "query" not in self._frame.columns:
            return None
        cached = getattr(self, "_texts_cache", None)
        if cached is None or cached[0] is not self._frame:
            perm, pair_off = self._csr()
            first_rows = pair_off[:-1] if perm is None else perm[pair_off[:-1]]  # every query number has >= 1 row
            cached = (self._frame, self._frame["query"].to_numpy()[first_rows].tolist())
            self._texts_cache = cached
        return cached[1]

    def __init__(
        self,
        df: pd.DataFrame,
        name: str | None = None,
        queries: Mapping[str, str] | None = None,
        dtype: np.dtype = np.dtype(np.float32),
        copy: bool = True,
        is_sorted: bool = False,
    ) -> None:
        """Build a ranking from a frame with columns ``q_id``, ``id``, ``score`` [, ``query``].

        Rows whose score is NaN are dropped.  Raises ``ValueError`` for a repeated
        (query, id) pair and for incomplete ``queries`` (ranking.py:67-121).
        """
        self.name = name
        if self._init_from_columns(df, queries, np.dtype(dtype), copy, is_sorted):
            return
        if not getattr(df, "_ffx_pairs_unique", False) and df.duplicated(subset=_KEY).any():
            raise ValueError("Only one score per query-document/passage pair is allowed.")

        keep = _KEY + ["score"] + (["query"] if "query" in df.columns else [])
        frame = df.loc[:, keep]
        if getattr(df, "_ffx_pairs_unique", False):
            # derived from an existing Ranking: ids and queries are known to be present, only a
            # score can be missing (early stopping leaves unscored candidates as NaN)
            scored = frame["score"].notna().to_numpy()
            if not scored.all():
                frame = frame[scored]
        else:
            frame = frame.dropna()
        if copy:
            frame = frame.copy()
        wanted = {"score": dtype, "q_id": str, "id": str}
        for col, dt in wanted.items():
            if frame[col].dtype != dt:
                frame[col] = frame[col].astype(dt)
        if not is_sorted:
            frame = _sorted_frame(frame)
        self._df = frame.reset_index(drop=True)
        # one pass gives both the set of query ids and every row's query number (order of first
        # appearance, base.py:418-421), which Index.__call__ needs again
        self._q_no, q_ids = pd.factorize(self._df["q_id"])
        self._q_order = list(q_ids)
        self._q_ids = set(self._q_order)
        if queries is not None:
            self._df = _with_queries(self._df, queries)

    def _init_from_columns(self, df: pd.DataFrame, queries, dtype: np.dtype, copy: bool, is_sorted: bool) -> bool:
        """The constructor for the ordinary large frame -- string ids, numeric scores, nothing missing but possibly
        some scores -- without a pandas hash of a Python string per row: every text column is numbered natively
        (``_coded_text_column``: Arrow-backed and categorical columns from their buffers, object columns through the
        column's object POINTERS first, then only the distinct strings), and the duplicate check, the NaN drop, the sort
        and the per-row query numbers all work on those integers (``_build_coded``).  The result is kept in the coded
        form (query number, id code, score per row + the two string tables); the frame of strings is built if and when
        somebody asks for it.  Returns False (nothing done) for anything else -- a non-string cell, a query id with two
        different texts, no native library -- and the general code decides."""
        from pandas.api.types import is_bool_dtype, is_numeric_dtype

        if len(df) < 2048 or any(c not in df.columns for c in (*_KEY, "score")):
            return False
        s_col = df["score"]
        if not is_numeric_dtype(s_col.dtype) or is_bool_dtype(s_col.dtype) or not isinstance(s_col.dtype, np.dtype):
            return False
        native = _native_or_none()
        if native is None:
            return False
        coded = {}
        laps = BUILD_TIMINGS
        laps.clear()
        for c in [*_KEY] + (["query"] if "query" in df.columns else []):
            t0 = perf_counter()
            coded[c] = _coded_text_column(native, df[c])  # (None: missing values, numbers, ...: the general path knows what to do)
            laps[f"column_{c}_numbered_s"] = perf_counter() - t0
            if coded[c] is None:
                return False
        return self._build_coded(native, coded["q_id"], coded["id"], s_col.to_numpy(), queries, coded.get("query"), dtype, is_sorted,
                                 getattr(df, "_ffx_pairs_unique", False))

    def _build_coded(self, native, q_col, id_col, score: np.ndarray, queries, text_col, dtype: np.dtype, is_sorted: bool, pairs_unique: bool) -> bool:
        """The coded form from numbered columns: ``q_col`` / ``id_col`` / ``text_col`` = (int32 code of every row, the
        distinct strings by code).  What the reference's constructor does with strings (ranking.py:95-117) -- repeated
        pairs refused, rows without a score dropped, rows ordered by query id descending and score descending, queries
        numbered in order of first appearance -- on integers.  A run that stands grouped by query (any run file, any frame
        made from one) is ordered query by query by host threads, scores and id codes moved along in the same pass
        (``ffx_host_rank_blocks``), and looked through for repeated pairs block by block.  False: one query id with two
        different texts (nothing done; the frame path decides)."""
        laps, t0 = BUILD_TIMINGS, perf_counter()
        q_code, q_names = q_col
        id_code, id_names = id_col
        n_q = len(q_names)
        q_names = np.asarray(q_names)
        text_of_q = t_names = None
        if text_col is not None:
            t_code, t_names = text_col
            text_of_q = np.full(n_q, -1, dtype=np.int32)
            text_of_q[q_code] = t_code
            if not np.array_equal(text_of_q[q_code], t_code):
                return False  # one query id with two different texts: leave it to the frame path
        if score.dtype.kind == "f":
            missing = np.isnan(score)
            if missing.any():  # rows without a score are dropped (ranking.py:103)
                rows = np.flatnonzero(~missing)
                q_code, id_code, score = q_code[rows], id_code[rows], score[rows]
        score = np.ascontiguousarray(score.astype(dtype, copy=False))
        n = len(q_code)
        starts = np.concatenate([[0], np.flatnonzero(q_code[1:] != q_code[:-1]) + 1]) if n else np.zeros(0, dtype=np.int64)
        run_q = q_code[starts]
        laps["texts_checked_scores_cast_blocks_found_s"] = perf_counter() - t0
        t0 = perf_counter()
        rank_of = np.empty(n_q, dtype=np.int64)  # position of every query id in descending order (ranking.py:115-117)
        rank_of[np.argsort(q_names)[::-1]] = np.arange(n_q)
        block_off = None
        if n and np.unique(run_q).shape[0] == run_q.shape[0] and score.dtype == np.float32:
            # one block per query: ordered inside by threads, the blocks laid out by rank, repeated pairs looked for block by block
            off = np.concatenate([starts, [n]]).astype(np.int64)
            id_code = np.ascontiguousarray(id_code, dtype=np.int32)
            if is_sorted:
                block_off, firsts = off, run_q
                score, id_codes = native.page_locked_copy(score), native.page_locked_copy(id_code)
            else:
                by_rank = np.argsort(rank_of[run_q], kind="stable")
                block_off = np.zeros(len(by_rank) + 1, dtype=np.int64)
                np.cumsum(np.diff(off)[by_rank], out=block_off[1:])
                dst = np.empty(len(by_rank), dtype=np.int64)
                dst[by_rank] = block_off[:-1]  # where block q lands
                score, id_codes = native.rank_blocks(score, id_code, off, dst)
                firsts = run_q[by_rank]
            laps["queries_ordered_columns_moved_s"] = perf_counter() - t0
            t0 = perf_counter()
            if not pairs_unique and native.repeated_in_groups(id_codes, block_off, len(id_names)) >= 0:
                raise ValueError("Only one score per query-document/passage pair is allowed.")
            laps["repeated_pairs_looked_for_s"] = perf_counter() - t0
            t0 = perf_counter()
            q_no = native.run_numbers(block_off)
            laps["rows_numbered_by_query_s"] = perf_counter() - t0
        else:
            if not pairs_unique:
                key = q_code.astype(np.int64) * max(1, len(id_names)) + id_code
                key.sort()
                if (key[1:] == key[:-1]).any():
                    raise ValueError("Only one score per query-document/passage pair is allowed.")
                del key
            codes = q_code
            if not is_sorted:
                order = np.ascontiguousarray(_order_by_query_then_score(rank_of[q_code], score), dtype=np.int32)
                score = native.take32(score, order) if score.dtype.itemsize == 4 else score[order]
                codes, id_code = native.take32(q_code, order), native.take32(np.ascontiguousarray(id_code, dtype=np.int32), order)
            # query numbers: order of first appearance in the final row order (base.py:418-421)
            starts = np.concatenate([[0], np.flatnonzero(codes[1:] != codes[:-1]) + 1]) if n else np.zeros(0, dtype=np.int64)
            if n and np.unique(codes[starts]).shape[0] == starts.shape[0]:
                firsts = codes[starts]  # rows of a query adjacent: the numbers follow the blocks
                block_off = np.concatenate([starts, [n]]).astype(np.int64)
                q_no = native.run_numbers(block_off)
            else:
                q_no, firsts = pd.factorize(codes)
                q_no = q_no.astype(np.int64, copy=False)
            # what an index uploads on every call on this ranking -- first-stage scores and id codes -- lives in page-locked blocks
            # when it is large and a device is there (native.page_locked_copy: DMA at link rate instead of a staged copy per call)
            id_codes = native.page_locked_copy(np.ascontiguousarray(id_code, dtype=np.int32))
            if score.dtype == np.float32:
                score = native.page_locked_copy(score)
        q_order = [q_names[i] for i in firsts]
        if queries is not None:
            if any(q not in queries for q in q_order):
                raise ValueError("Queries are incomplete.")
            q_texts = [queries[q] for q in q_order]
        else:
            q_texts = None if text_of_q is None else [t_names[text_of_q[i]] for i in firsts]
        self._frame = None
        self._id_cache = None
        self._coded = {"q_no": q_no, "q_order": q_order, "id_codes": id_codes, "id_names": id_names,
                       "score": score, "q_texts": q_texts, "id_blob": None}
        self._q_no, self._q_order = q_no, q_order
        self._q_ids = set(q_order)
        if block_off is not None:  # the CSR by query an index asks for (_csr) is known already: no pass over the rows later
            self._csr_cache = (q_no, None, block_off)
        return True

    @classmethod
    def from_columns(cls, q_ids, ids, scores, name: "str | None" = None, queries: "Mapping[str, str] | None" = None,
                     dtype: np.dtype = np.dtype(np.float32), is_sorted: bool = False) -> "Ranking":
        """A ranking from COLUMNS -- one entry per (query, candidate) row -- without a Python object per row: what
        ``Ranking(pd.DataFrame({"q_id": q_ids, "id": ids, "score": scores}), ...)`` gives (same checks, same order,
        same values; ranking.py:67-121), for runs of tens of millions of rows.  ``q_ids`` / ``ids`` may each be

        * a pyarrow ``string`` / ``large_string`` array or chunked array, or a pandas Series backed by one
          (``dtype="string[pyarrow]"``, ``pd.ArrowDtype(pa.string())``): numbered from its offsets and data buffers;
        * a NumPy array of fixed-width bytes (``S``) or ASCII text (``U``);
        * pre-coded: a ``(codes, names)`` pair -- integer code of every row, distinct strings by code -- or a pandas
          ``Categorical``;
        * an object array / list of ``str`` (numbered through the objects' addresses first).

        ``scores``: a numeric array; rows whose score is NaN are dropped.  Raises ``ValueError`` for a repeated (query, id)
        pair and for incomplete ``queries``; ``TypeError`` for a column that is none of the above or holds missing values."""
        native = _native_or_none()
        if native is None:
            raise ImportError("Ranking.from_columns needs libffx (build it: python -c 'import __graft_entry__ as g; g.build()')")
        scores = np.asarray(scores)
        if scores.ndim != 1 or scores.dtype.kind not in "fiu":
            raise TypeError("scores must be a one-dimensional numeric array")
        cols = []
        BUILD_TIMINGS.clear()
        for label, col in (("q_ids", q_ids), ("ids", ids)):
            t0 = perf_counter()
            got = _coded_text_column(native, col)
            BUILD_TIMINGS[f"column_{label}_numbered_s"] = perf_counter() - t0
            if got is None:
                raise TypeError(f"{label}: not a column of strings without missing values (see Ranking.from_columns)")
            if len(got[0]) != scores.shape[0]:
                raise ValueError("q_ids, ids and scores must have one entry per row each")
            cols.append(got)
        out = cls.__new__(cls)
        out.name = name
        if scores.shape[0] == 0:
            out._df = pd.DataFrame({"q_id": np.empty(0, dtype=object), "id": np.empty(0, dtype=object), "score": np.empty(0, dtype=np.dtype(dtype))})
            out._q_no, out._q_order, out._q_ids = np.zeros(0, dtype=np.int64), [], set()
            return out
        built = out._build_coded(native, cols[0], cols[1], scores, queries, None, np.dtype(dtype), is_sorted, False)
        assert built
        return out

    def columns(self) -> dict:
        """This ranking as columns, without a Python object per row: ``{"q_codes", "q_names", "id_codes", "id_names",
        "score", "pair_off"}`` -- the query of row i is ``q_names[q_codes[i]]``, its candidate ``id_names[id_codes[i]]``
        (``id_names``: a :class:`~fast_forward_amd.names.Names`, the distinct ids as one buffer; ``np.asarray`` makes
        strings of it); ``pair_off``: the rows of query number q are ``pair_off[q] .. pair_off[q+1]`` when the rows stand
        grouped by query (they do for every ranking a constructor or an index made), else None.  Rows in this ranking's
        order (query id descending, score descending).  The arrays are the ranking's own: do not write to them."""
        if self._coded is None:
            codes, names = self._id_coding()
            packed = native_packed = None
            native = _native_or_none()
            if native is not None:
                native_packed = native._packed(names)
            id_names = Names(native_packed[0], native_packed[1], objects=names) if native_packed is not None else names
            del packed
            score = self._frame["score"].to_numpy()
        else:
            codes, id_names, score = self._coded["id_codes"], self._coded["id_names"], self._coded["score"]
        perm, pair_off = self._csr() if len(self._q_no) else (None, np.zeros(len(self._q_order) + 1, dtype=np.int64))
        return {"q_codes": self._q_no, "q_names": list(self._q_order), "id_codes": codes, "id_names": id_names, "score": score,
                "pair_off": pair_off if perm is None else None}

    @classmethod
    def _assemble(cls, columns: Mapping[str, np.ndarray], name: str | None, q_no: np.ndarray, q_order: list[str]) -> "Ranking":
        """Ranking from finished columns, with no scan at all.  For results derived on the device
        from an existing Ranking: rows already in final order (q_id descending, score descending),
        pairs unique, no missing values; ``q_no`` / ``q_order`` as the constructor computes them."""
        out = cls.__new__(cls)
        out.name = name
        out._df = pd.DataFrame(dict(columns), copy=False)
        out._q_no, out._q_order = q_no, list(q_order)
        out._q_ids = set(out._q_order)
        return out

    def _numbering(self) -> tuple[np.ndarray, list[str]]:
        """(query number of every row, query ids in that numbering): order of first appearance."""
        return self._q_no, self._q_order

    # ---- container protocol ------------------------------------------------------------
    @property
    def has_queries(self) -> bool:
        """Whether query strings are attached."""
        if self._coded is not None:
            return self._coded["q_texts"] is not None
        return "query" in self._df.columns

    @property
    def q_ids(self) -> set[str]:
        """Ids of the queries that have at least one scored candidate."""
        return self._q_ids

    def __getitem__(self, q_id: str) -> dict[str, float]:
        rows = self._df[self._df["q_id"] == q_id]
        return dict(zip(rows["id"], rows["score"]))

    def __len__(self) -> int:
        return len(self._q_ids)

    def __iter__(self) -> Iterator[str]:
        yield from self._q_ids

    def __contains__(self, key: object) -> bool:
        return key in self._q_ids

    def __eq__(self, o: object) -> bool:
        """Same (q_id, id, score) triples, in any order (ranking.py:171-186)."""
        if not isinstance(o, Ranking):
            return False
        cols = _KEY + ["score"]
        mine = self._df.sort_values(_KEY).reset_index(drop=True)[cols]
        theirs = o._df.sort_values(_KEY).reset_index(drop=True)[cols]
        return mine.equals(theirs)

    __hash__ = None  # mutable container, like the reference

    def __repr__(self) -> str:
        return repr(self._df)

    # ---- score algebra ---------------------------------------------------------------------
    def _joined(self, other: "Ranking", left: pd.DataFrame | None = None, right: pd.DataFrame | None = None) -> pd.DataFrame:
        """Outer join on (q_id, id); a side that lacks the pair contributes 0."""
        a = self._df if left is None else left
        b = other._df if right is None else right
        return a.merge(b, on=_KEY, how="outer", suffixes=(None, "_other")).fillna(0)

    def _derived(self, df: pd.DataFrame, is_sorted: bool, copy: bool = False) -> "Ranking":
        return Ranking(df, name=self.name, dtype=self._df.dtypes["score"], copy=copy, is_sorted=is_sorted)

    @staticmethod
    def _trusted(df: pd.DataFrame) -> pd.DataFrame:
        """Mark a frame whose (q_id, id) pairs are a subset of an existing Ranking's: the
        constructor then skips the duplicate scan (a string-pair hash of every row)."""
        object.__setattr__(df, "_ffx_pairs_unique", True)
        return df

    def _combined(self, other: "Ranking", fn, left: pd.DataFrame | None = None, right: pd.DataFrame | None = None) -> "Ranking":
        """``fn(score, score_other)`` over the outer join on (q_id, id) -- what ``_joined`` + the
        constructor compute, without hashing a string pair per row: queries are numbered through the
        cached per-row query numbers, ids through the two rankings' id codes (shared as they are when
        ``other`` came out of ``index(self)``; otherwise the two tables of DISTINCT ids are merged), and
        the join runs on the resulting integer keys.  Same rows, scores and order (exact ties: by id,
        as the sorted outer merge leaves them).  The result is coded whenever every row has its query
        text on the side that provides it (always, when both sides hold the same pairs)."""
        sa = self._scores() if left is None else left["score"].to_numpy()
        sb = other._scores() if right is None else right["score"].to_numpy()
        na, nb = len(sa), len(sb)
        if na + nb == 0:
            return None
        qa, names_a = self._numbering()
        qb, names_b = other._numbering()
        code_of = {q: i for i, q in enumerate(names_a)}
        for q in names_b:
            code_of.setdefault(q, len(code_of))
        names = list(code_of)
        gb = np.fromiter((code_of[q] for q in names_b), dtype=np.int64, count=len(names_b))[qb] if nb else np.zeros(0, dtype=np.int64)
        ca, ids_a = self._id_coding()
        cb, ids_b = other._id_coding()
        native = _native_or_none() if na + nb >= (1 << 16) else None
        if ids_a is ids_b:
            id_names, code_a, code_b = ids_a, ca, cb
        else:  # merge the two tables of DISTINCT ids (not the rows)
            merged = None
            if native is not None:  # numbered by threads from one joined buffer (no hash insert per string in Python)
                if isinstance(ids_a, Names) and isinstance(ids_b, Names):
                    packed = Names.joined(ids_a, ids_b).packed
                else:
                    packed = native._packed(np.concatenate([np.asarray(ids_a), np.asarray(ids_b)]))
                if packed is not None:
                    merged, names_blob, n_names = native.code_strings(*packed)
                    id_names = Names(names_blob, n_names)
            if merged is None:
                merged, uniques = pd.factorize(np.concatenate([np.asarray(ids_a), np.asarray(ids_b)]))
                id_names = np.asarray(uniques, dtype=object)
            code_a, code_b = merged[:len(ids_a)][ca], merged[len(ids_a):][cb]
        n_all = len(names)
        joined = None
        if native is not None:
            # both sides grouped by query (always, for rankings the constructor sorted): per-query sort + merge by threads
            pa, off_a = self._csr()
            pb, off_b_own = other._csr()
            if pa is None and pb is None:
                cnt_a = np.zeros(n_all, dtype=np.int64)
                cnt_a[:len(names_a)] = np.diff(off_a)
                g_of_b = np.fromiter((code_of[q] for q in names_b), dtype=np.int64, count=len(names_b))
                cnt_b = np.zeros(n_all, dtype=np.int64)
                cnt_b[g_of_b] = np.diff(off_b_own)
                # B's blocks re-arranged into the common query numbering
                start_b = np.zeros(n_all, dtype=np.int64)
                start_b[g_of_b] = off_b_own[:-1]
                goff_a = np.zeros(n_all + 1, dtype=np.int64)
                np.cumsum(cnt_a, out=goff_a[1:])
                goff_b = np.zeros(n_all + 1, dtype=np.int64)
                np.cumsum(cnt_b, out=goff_b[1:])
                rows_b = (np.arange(nb, dtype=np.int64) + np.repeat(start_b - goff_b[:-1], cnt_b)).astype(np.int32) if nb else np.zeros(0, dtype=np.int32)
                in_common_order = bool((np.diff(rows_b) == 1).all()) if nb > 1 else True
                cb_g = np.ascontiguousarray(code_b, dtype=np.int32) if in_common_order else native.take32(np.ascontiguousarray(code_b, dtype=np.int32), rows_b)
                u_off, u_code, u_a, u_b = native.outer_join(goff_a, np.ascontiguousarray(code_a, dtype=np.int32), goff_b, cb_g)
                if not in_common_order:  # rows of B as B numbers them
                    u_b = np.where(u_b >= 0, rows_b[np.maximum(u_b, 0)], -1).astype(np.int32)
                joined = (np.repeat(np.arange(n_all, dtype=np.int64), np.diff(u_off)), u_code, u_a, u_b)
        if joined is None:  # the NumPy statement of the same join
            width = max(1, len(id_names))
            keys, where = np.unique(np.concatenate([np.asarray(qa, dtype=np.int64) * width + np.asarray(code_a, dtype=np.int64),
                                                    gb * width + np.asarray(code_b, dtype=np.int64)]), return_inverse=True)
            u_a = np.full(keys.shape[0], -1, dtype=np.int64)
            u_a[where[:na]] = np.arange(na)
            u_b = np.full(keys.shape[0], -1, dtype=np.int64)
            u_b[where[na:]] = np.arange(nb)
            joined = (keys // width, (keys % width).astype(np.int32), u_a, u_b)
        q_of, u_code, u_a, u_b = joined
        n = len(u_code)
        has_a, has_b = u_a >= 0, u_b >= 0
        va = np.where(has_a, sa[np.maximum(u_a, 0)], sa.dtype.type(0)) if na else np.zeros(n, dtype=sa.dtype)
        vb = np.where(has_b, sb[np.maximum(u_b, 0)], sb.dtype.type(0)) if nb else np.zeros(n, dtype=sb.dtype)
        score = np.asarray(fn(va, vb)).astype(self._score_dtype(), copy=False)
        if np.isnan(score).any():
            return None  # rows would be dropped: leave it to the general path
        # final order: q_id descending, score descending, exact ties by id ascending
        rank_of = np.empty(len(names), dtype=np.int64)
        rank_of[np.argsort(np.array(names, dtype=object))[::-1]] = np.arange(len(names))
        q_rank = rank_of[q_of]
        order = _order_by_query_then_score(q_rank, score)
        id_code = np.asarray(u_code)[order].astype(np.int32)
        s_o, q_o = score[order], q_rank[order]
        tied = np.flatnonzero((s_o[1:] == s_o[:-1]) & (q_o[1:] == q_o[:-1]))
        if tied.size:
            starts = tied[np.concatenate([[True], np.diff(tied) > 1])]
            for lo in starts.tolist():
                hi = lo + 1
                while hi < n and s_o[hi] == s_o[lo] and q_o[hi] == q_o[lo]:
                    hi += 1
                fix = np.argsort(id_names[id_code[lo:hi]], kind="stable")
                order[lo:hi] = order[lo:hi][fix]
                id_code[lo:hi] = id_code[lo:hi][fix]
        q_names = np.array(names, dtype=object)
        # numbering of the result: queries appear in rank order
        present = np.flatnonzero(np.bincount(q_o, minlength=len(names)))
        renumber = np.full(len(names), -1, dtype=np.int64)
        renumber[present] = np.arange(present.shape[0])
        by_rank = np.empty(len(names), dtype=object)
        by_rank[rank_of] = q_names
        q_order = by_rank[present].tolist()
        # the query text comes from this ranking if it has any, else from `other`; a pair the
        # chosen side lacks gets the merge's fill value 0
        side = (self, u_a) if self.has_queries else (other, u_b) if other.has_queries else None
        if side is None or bool((side[1] >= 0).all()):
            q_texts = None
            if side is not None:  # every row has its text on that side: one text per query
                text_of = dict(zip(side[0]._numbering()[1], side[0]._query_texts()))
                q_texts = [text_of[q] for q in q_order]
            return Ranking._assemble_coded(self.name, renumber[q_o], q_order, id_code, id_names, s_o, q_texts)
        cols = {"q_id": q_names[q_of[order]], "id": id_names[id_code], "score": s_o}
        src = np.asarray(side[1])[order]
        query = np.zeros(n, dtype=object)
        have = src >= 0
        query[have] = side[0]._df["query"].to_numpy()[src[have]]
        cols["query"] = query
        return Ranking._assemble(cols, self.name, renumber[q_o], q_order)

    def __add__(self, o: "Ranking | float") -> "Ranking":
        if isinstance(o, Ranking):
            fast = self._combined(o, lambda x, y: x + y)
            if fast is not None:
                return fast
            both = self._joined(o)
            both["score"] = both["score"] + both["score_other"]
            return self._derived(both, is_sorted=False)
        if isinstance(o, (int, float)):
            shifted = self._df.copy()
            shifted["score"] += o
            return self._derived(shifted, is_sorted=True)
        return NotImplemented

    __radd__ = __add__

    def __mul__(self, o: float) -> "Ranking":
        if not isinstance(o, (int, float)):
            return NotImplemented
        scaled = self._df.copy()
        scaled["score"] *= o
        return self._derived(scaled, is_sorted=True)

    __rmul__ = __mul__

    def _interpolated_with_own_dense_scores(self, other: "Ranking", alpha: float) -> "Ranking | None":
        """``self.interpolate(index(self), alpha)`` -- the reference's documented pipeline -- without a join:
        a ranking that came out of ``Index.__call__(self)`` remembers which row of ``self`` each of its rows
        is (``_source_rows``), so the dense scores are put back in place by one scatter, combined with the same
        float32 expression as the general path, and every query is re-ordered by host threads
        (``ffx_host_order``: score descending, exact ties by first-stage position -- the order
        ``Index.rerank`` returns; the general path breaks exact ties by id, the reference leaves them
        unspecified).  None when this is not that case."""
        link = getattr(other, "_source_rows", None)
        if link is None or link[0]() is not self or len(link[1]) != len(self._q_no):
            return None
        perm, pair_off = self._csr()
        q_order = self._q_order
        if perm is not None or any(q_order[i] <= q_order[i + 1] for i in range(len(q_order) - 1)):
            return None  # rows not in the constructor's order: the general path sorts them
        mine = self._scores()
        if mine.dtype != np.float32 or other._scores().dtype != np.float32:
            return None
        try:
            from fast_forward_amd import native

            native.lib()
        except (ImportError, OSError):
            return None
        # A large run whose dense scores are still in HBM where the index left them when it made them (the token names that
        # very call): the device interpolates with THIS alpha and orders (Index._interpolate_kept) -- faster than the host
        # orders 35 M scores; same bits, pinned by test_rerank_equals_call_then_interpolate.  Once another call has displaced
        # them (or the call was batched and kept nothing), the host path below combines `other`'s OWN scores: interpolate is
        # arithmetic on the two rankings (ranking.py:293-326) -- it never encodes a query or scores a candidate again, so an
        # index whose encoder, summation or contents changed in between cannot change what it returns.
        origin = getattr(other, "_origin", None)
        index = origin[0]() if origin is not None else None
        if (index is not None and isinstance(alpha, (int, float)) and len(mine) >= (1 << 20) and self.has_queries and origin[1] == index.mode
                and origin[2] == getattr(getattr(index, "_ids", None), "version", None) and hasattr(index, "_score_ranked")):
            token = origin[3] if len(origin) > 3 else 0
            if token and index._kept_token() == token:  # the dense scores of that very call are still in HBM: no second gather
                return index._interpolate_kept(self, token, alpha)
        if isinstance(alpha, (int, float)) and len(mine) >= (1 << 18) and link[1].dtype == np.int32:
            # one threaded pass: scatter back + `alpha * mine + (1 - alpha) * dense` in float32, as below
            score, any_nan = native.host_interpolate(mine, other._scores(), link[1], alpha)
            if any_nan:
                return None
        else:
            dense = np.empty_like(mine)
            dense[link[1]] = other._scores()
            score = alpha * mine + (1 - alpha) * dense  # a Python-float weight on a float32 array multiplies in float32 (ranking.py:319)
            if score.dtype != np.float32 or np.isnan(score).any():
                return None
        order = native.host_order(score, pair_off)
        codes, names = self._id_coding()
        texts = self._query_texts() if self.has_queries else other._query_texts()
        return Ranking._assemble_coded(self.name, self._q_no, q_order, native.take32(codes, order), names, native.take32(score, order), texts)

    def interpolate(self, other: "Ranking", alpha: float, normalize: bool = False) -> "Ranking":
        """``alpha * self + (1 - alpha) * other`` per (query, id) pair (ranking.py:293-326).

        When ``other`` was produced by ``Index.__call__(self)`` on the device, prefer
        ``Index.__call__(..., alpha=...)`` / ``Index.rerank`` which fuse this step into the
        scoring kernel's epilogue; this method is the general host path and gives the same bits.
        """
        if not normalize:
            direct = self._interpolated_with_own_dense_scores(other, alpha)
            if direct is not None:
                return direct
        left = _scaled_to_unit_range(self._df) if normalize else None
        right = _scaled_to_unit_range(other._df) if normalize else None
        # a Python-float weight on a float32 array multiplies in float32 (ranking.py:319)
        fast = self._combined(other, lambda x, y: alpha * x + (1 - alpha) * y, left, right)
        if fast is not None:
            return fast
        both = self._joined(other, left, right)
        # a Python-float weight on a float32 column multiplies in float32 (ranking.py:319)
        both["score"] = alpha * both["score"] + (1 - alpha) * both["score_other"]
        return self._derived(both, is_sorted=False)

    def normalize(self) -> "Ranking":
        """Min-max scale all scores to [0, 1] (constant scores -> 0)."""
        return self._derived(_scaled_to_unit_range(self._df), is_sorted=True)

    def rr_scores(self, k: int = 60) -> "Ranking":
        """Reciprocal-rank scores ``1 / (rank + k)`` (ranking.py:328-346)."""
        out = self._df.copy()
        out["score"] = 1 / (_ranks_within_query(out) + k)
        return self._derived(out, is_sorted=True)

    # ---- reshaping -------------------------------------------------------------------------
    def attach_queries(self, queries: Mapping[str, str]) -> "Ranking":
        """Copy of this ranking with query strings attached (``ValueError`` if incomplete)."""
        if any(q not in queries for q in self._q_order):
            raise ValueError("Queries are incomplete.")
        if self._coded is not None:  # stays coded: only the texts are new
            c = self._coded
            out = Ranking._assemble_coded(self.name, c["q_no"], c["q_order"], c["id_codes"], c["id_names"], c["score"],  # (shared: a Ranking never changes its arrays)
                                          [queries[q] for q in self._q_order], c.get("id_blob"))
            cached = getattr(self, "_csr_cache", None)
            if cached is not None and cached[0] is self._q_no:
                out._csr_cache = cached
            return out
        # rows, order and pairs are this ranking's own: nothing to re-check, the texts follow the
        # per-row query numbers
        cols = {c: self._df[c].to_numpy().copy() for c in (*_KEY, "score")}
        texts = np.empty(len(self._q_order), dtype=object)
        texts[:] = [queries[q] for q in self._q_order]
        cols["query"] = texts[self._q_no] if len(self._q_order) else np.empty(0, dtype=object)
        return Ranking._assemble(cols, self.name, self._q_no, self._q_order)

    def cut(self, cutoff: int) -> "Ranking":
        """Keep the ``cutoff`` best candidates of every query (frame is sorted by score)."""
        q_no = self._q_no
        if len(q_no) and (np.diff(q_no) >= 0).all():
            # rows of a query are adjacent (always, for a sorted frame): rank = position inside the run
            starts = np.zeros(len(self._q_order) + 1, dtype=np.int64)
            np.cumsum(np.bincount(q_no, minlength=len(self._q_order)), out=starts[1:])
            if self._coded is not None and cutoff >= int(np.diff(starts).max()):
                # nothing to cut (the usual `.cut(k)` on a run of depth k): the same rows; the arrays are shared,
                # a Ranking never changes them
                c = self._coded
                return Ranking._assemble_coded(self.name, c["q_no"], c["q_order"], c["id_codes"], c["id_names"], c["score"], c["q_texts"], c.get("id_blob"))
            keep = np.flatnonzero(np.arange(len(q_no)) - starts[q_no] < cutoff)
            present = np.flatnonzero(np.bincount(q_no[keep], minlength=len(self._q_order)))
            renumber = np.full(len(self._q_order), -1, dtype=np.int64)
            renumber[present] = np.arange(present.shape[0])
            if self._coded is not None:  # stays coded (the name table may keep names no row uses any more)
                c = self._coded
                texts = None if c["q_texts"] is None else [c["q_texts"][i] for i in present]
                return Ranking._assemble_coded(self.name, renumber[q_no[keep]], [self._q_order[i] for i in present], c["id_codes"][keep],
                                               c["id_names"], c["score"][keep], texts, c.get("id_blob"))
            cols = {c: self._df[c].to_numpy()[keep] for c in self._df.columns}
            return Ranking._assemble(cols, self.name, renumber[q_no[keep]], [self._q_order[i] for i in present])
        top = self._df[_ranks_within_query(self._df) <= cutoff]
        return self._derived(top.reset_index(drop=True), is_sorted=True, copy=True)

    # ---- I/O -----------------------------------------------------------------------------------
    def save(self, target: Path) -> None:
        """Write a TREC run file: ``q_id Q0 id rank score name`` (ranking.py:348-366)."""
        target = Path(target)
        if self._coded is not None and self._score_dtype() == np.float32 and self._save_native_coded(target):
            return
        if len(self._df) >= 4096 and self._df["score"].dtype == np.float32 and self._save_native(target):
            return
        out = self._df[_KEY + ["score"]].copy()
        out["rank"] = _ranks_within_query(out)
        out["q0"] = "Q0"
        out["name"] = str(self.name)
        target.parent.mkdir(parents=True, exist_ok=True)
        out.to_csv(target, sep="\t", columns=["q_id", "q0", "id", "rank", "score", "name"], index=False, header=False)

    @classmethod
    def _from_file_native(cls, f: Path, queries, dtype: np.dtype) -> "Ranking | None":
        """``from_file`` through libffx's threaded parser, for files where that provably gives what
        ``pandas.read_csv`` + the constructor give: float32 scores, six fields per line, no quotes,
        id columns that pandas would leave as the strings they are (text, or integers that print
        back unchanged) with none of its NA words, a textual run name.  Anything else -- and any
        parse problem -- returns None and the pandas route decides."""
        if dtype != np.float32:
            return None
        try:
            if f.stat().st_size < (1 << 18):
                return None
            from fast_forward_amd import native

            native.lib()
            run = native.RunFile(f, None)
        except (ImportError, OSError, ValueError, IndexError):
            return None
        try:
            q_flags, id_flags, misc = run.flags()

            def strings_survive(flags: int) -> bool:
                if flags & native.TOKEN_NA:
                    return False
                return bool(flags & native.TOKEN_TEXT) or flags == native.TOKEN_INT

            name = run.tag
            if misc or not name or not strings_survive(q_flags) or not strings_survive(id_flags):
                return None
            if not _is_plain_text(name):
                return None
            q_names = np.array(run.query_ids(), dtype=object)
            if run.n_pairs >= _CODED_FROM_FILE_ROWS:
                pair_off, codes, scores = run.export()
                if not np.isnan(scores).any():
                    return cls._from_parsed_run_coded(run, native, name, pair_off, codes, scores, q_names, queries)
            pair_off, scores = run.scores()
            ids = np.array(run.id_strings(), dtype=object)
        finally:
            run.close()
        counts = np.diff(pair_off)
        q_of = np.repeat(np.arange(len(q_names)), counts)
        keep = ~np.isnan(scores)
        if not keep.all():  # dropna (ranking.py:103)
            q_of, ids, scores = q_of[keep], ids[keep], scores[keep]
        # q_id descending, score descending, exact ties in file order
        rank_of = np.empty(len(q_names), dtype=np.int64)
        rank_of[np.argsort(q_names)[::-1]] = np.arange(len(q_names))
        q_rank = rank_of[q_of]
        order = np.lexsort((-scores.astype(np.float64), q_rank))
        q_sorted = q_rank[order]
        present = np.flatnonzero(np.bincount(q_sorted, minlength=len(q_names)))
        renumber = np.full(len(q_names), -1, dtype=np.int64)
        renumber[present] = np.arange(present.shape[0])
        by_rank = np.empty(len(q_names), dtype=object)
        by_rank[rank_of] = q_names
        out = cls._assemble({"q_id": q_names[q_of[order]], "id": ids[order], "score": scores[order]}, name,
                            renumber[q_sorted], by_rank[present].tolist())
        if queries is not None:
            out._df = _with_queries(out._df, queries)
        return out

    @classmethod
    def _from_parsed_run_coded(cls, run, native, name: str, pair_off: np.ndarray, codes: np.ndarray, scores: np.ndarray, q_names: np.ndarray,
                               queries) -> "Ranking":
        """A large run file straight into the coded form: the parser has numbered the candidate ids already (no string per
        row ever exists, the DISTINCT names stay one buffer: ``Names``), every query is ordered by host threads (score
        descending, exact ties in file order: what the constructor's stable sort gives) and its block written where the
        query stands in descending id order, codes and scores moved along in the same pass.  Same rows, order and values as
        the frame route (``test_from_file_coded_equals_the_frame_route``)."""
        from fast_forward_amd.names import Names

        id_names = Names(*run.names())
        counts = np.diff(pair_off)
        by_name = np.argsort(q_names)[::-1]                         # q_id descending (ranking.py:115-117)
        by_name = by_name[counts[by_name] > 0]
        new_off = np.zeros(len(by_name) + 1, dtype=np.int64)
        np.cumsum(counts[by_name], out=new_off[1:])
        dst = np.zeros(len(q_names), dtype=np.int64)                # where the block of (parser) query q starts; empty queries: anywhere
        dst[by_name] = new_off[:-1]
        score_o, code_o = native.rank_blocks(scores, codes, pair_off, dst)
        q_order = q_names[by_name].tolist()
        if queries is not None and any(q not in queries for q in q_order):
            raise ValueError("Queries are incomplete.")
        texts = None if queries is None else [queries[q] for q in q_order]
        out = cls._assemble_coded(name, native.run_numbers(new_off), q_order, code_o, id_names, score_o, texts)
        out._csr_cache = (out._q_no, None, new_off)
        return out

    def _save_native_coded(self, target: Path) -> bool:
        """``save`` of a coded ranking without growing its frame: codes and the name table go to libffx's
        formatter as they are (``ffx_run_save_coded``); False when that does not apply."""
        try:
            from fast_forward_amd import native

            native.lib()
        except (ImportError, OSError):
            return False
        c = self._coded
        if len(c["q_no"]) < 4096:
            return False
        perm, pair_off = self._csr()  # (known from the constructor for every ranking it made: no pass over the rows here)
        if perm is not None:
            return False
        target.parent.mkdir(parents=True, exist_ok=True)
        names = c["id_names"]
        if isinstance(names, Names):
            if not names.plain_for_csv():
                return False
            names = names.packed or (b"", 0)
        return native.save_run_coded(target, c["q_order"], pair_off, c["id_codes"], names, c["score"], str(self.name))

    def _save_native(self, target: Path) -> bool:
        """The same bytes through libffx's threaded formatter (pandas' to_csv spends seconds on a
        million rows); False when the library is not there or a field would need CSV quoting."""
        try:
            from fast_forward_amd import native

            native.lib()
        except (ImportError, OSError):
            return False
        q_no, q_order = self._numbering()
        if (np.diff(q_no) < 0).any():  # rows of a query not adjacent: ranks would differ
            return False
        pair_off = np.zeros(len(q_order) + 1, dtype=np.int64)
        np.cumsum(np.bincount(q_no, minlength=len(q_order)), out=pair_off[1:])
        target.parent.mkdir(parents=True, exist_ok=True)
        return native.save_run(target, q_order, pair_off, self._df["id"].to_numpy(), self._df["score"].to_numpy(), str(self.name))

    @classmethod
    def from_run(cls, run: Run, name: str | None = None, queries: Mapping[str, str] | None = None,
                 dtype: np.dtype = np.dtype(np.float32)) -> "Ranking":
        """From ``{q_id: {id: score}}`` (ranking.py:368-386)."""
        q_col, id_col, s_col = [], [], []
        for q_id, scored in run.items():
            q_col.extend([q_id] * len(scored))
            id_col.extend(scored.keys())
            s_col.extend(scored.values())
        df = pd.DataFrame({"q_id": pd.Series(q_col, dtype=object), "id": pd.Series(id_col, dtype=object),
                           "score": pd.Series(s_col, dtype=np.float64)})
        return cls(df, name=name, queries=queries, dtype=dtype, copy=False)

    @classmethod
    def from_file(cls, f: Path, queries: Mapping[str, str] | None = None,
                  dtype: np.dtype = np.dtype(np.float32)) -> "Ranking":
        """From a TREC run file; the method name is taken from its last column."""
        fast = cls._from_file_native(Path(f), queries, np.dtype(dtype))
        if fast is not None:
            return fast
        df = pd.read_csv(f, sep=r"\s+", skipinitialspace=True, header=None,
                         names=["q_id", "q0", "id", "rank", "score", "name"])
        return cls(df, name=df["name"][0], queries=queries, dtype=dtype, copy=False)
