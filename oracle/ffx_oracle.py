"""ctypes front-end of the CPU parity oracle + a NumPy port of the reference's hot lines.

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  The product package never imports this module.

Two restatements live here, both pinned against the reference by ``tests/golden``:

* ``libffx_oracle.so`` (``ffx_oracle.c``): scalar C, bit-exact model of the arithmetic
  (NumPy pairwise sum, pandas Kahan mean, two-multiply interpolation).
* ``numpy_port_*``: the same three NumPy operations the reference executes at
  ``src/fast_forward/index/base.py:301-303`` (gather, gather, ``np.sum(q*d, axis=1)``),
  used as the "reference-equivalent CPU path" that ``bench.py`` times beside the GPU.
* ``pandas_port_score``: the same path through frames of string ids, merges and a group-by, as a user of the
  reference experiences it (timed beside the NumPy port; checked against it and against the golden fixtures).
"""

from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "libffx_oracle.so"

MODE_PASSAGE, MODE_MAXP, MODE_FIRSTP, MODE_AVEP = 1, 2, 3, 4
OK, BAD_ARG, NOT_FOUND = 0, -1, -5


def build(force: bool = False) -> Path:
    """Compile the C oracle with the committed Makefile (gcc only)."""
    src = _HERE / "ffx_oracle.c"
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "-B", "libffx_oracle.so"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(_LIB_PATH))
        f32p, i32p, i64p = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        L.ffo_synth_rows.argtypes = [C.c_uint64, C.c_float, C.c_int64, C.c_int64, C.c_int64, f32p]
        L.ffo_synth_rows.restype = None
        L.ffo_synth_row_list.argtypes = [C.c_uint64, C.c_float, i64p, C.c_int64, C.c_int64, f32p]
        L.ffo_synth_row_list.restype = None
        L.ffo_dot_f32.argtypes = [f32p, f32p, C.c_int64]
        L.ffo_dot_f32.restype = C.c_float
        L.ffo_row_scores.argtypes = [f32p, f32p, C.c_int64, i32p, i32p, C.c_int64, f32p]
        L.ffo_row_scores.restype = None
        L.ffo_aggregate.argtypes = [f32p, i64p, C.c_int64, C.c_int, f32p]
        L.ffo_aggregate.restype = None
        L.ffo_interpolate.argtypes = [f32p, f32p, C.c_int64, C.c_double, f32p]
        L.ffo_interpolate.restype = None
        L.ffo_score.argtypes = [
            f32p, C.c_int64, C.c_int64, i64p, i32p, C.c_int64, i32p, C.c_int64,
            f32p, C.c_int64, i64p, i32p, C.c_int, f32p, C.c_double, f32p, f32p, i64p,
        ]
        L.ffo_score.restype = C.c_int
        L.ffo_order.argtypes = [f32p, C.c_int64, i64p, i64p]
        L.ffo_order.restype = C.c_int
        _lib = L
    return _lib


def _p(a: np.ndarray | None, ct):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def synth_rows(seed: int, scale: float, first_row: int, n_rows: int, dim: int) -> np.ndarray:
    out = np.empty((n_rows, dim), dtype=np.float32)
    lib().ffo_synth_rows(seed, scale, first_row, n_rows, dim, _p(out, C.c_float))
    return out


def synth_row_list(seed: int, scale: float, rows, dim: int) -> np.ndarray:
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    out = np.empty((rows.shape[0], dim), dtype=np.float32)
    lib().ffo_synth_row_list(seed, scale, _p(rows, C.c_int64), rows.shape[0], dim, _p(out, C.c_float))
    return out


def dot_f32(q, v) -> np.float32:
    q, v = _f32(q), _f32(v)
    assert q.shape == v.shape and q.ndim == 1
    return np.float32(lib().ffo_dot_f32(_p(q, C.c_float), _p(v, C.c_float), q.shape[0]))


def row_scores(queries, vectors, row_q, row_ids) -> np.ndarray:
    queries, vectors = _f32(queries), _f32(vectors)
    row_q = np.ascontiguousarray(row_q, dtype=np.int32)
    row_ids = np.ascontiguousarray(row_ids, dtype=np.int32)
    out = np.empty(row_ids.shape[0], dtype=np.float32)
    lib().ffo_row_scores(_p(queries, C.c_float), _p(vectors, C.c_float), vectors.shape[1],
                         _p(row_q, C.c_int32), _p(row_ids, C.c_int32), row_ids.shape[0], _p(out, C.c_float))
    return out


def aggregate(scores, seg_off, mode: int) -> np.ndarray:
    scores = _f32(scores)
    seg_off = np.ascontiguousarray(seg_off, dtype=np.int64)
    out = np.empty(seg_off.shape[0] - 1, dtype=np.float32)
    lib().ffo_aggregate(_p(scores, C.c_float), _p(seg_off, C.c_int64), out.shape[0], mode, _p(out, C.c_float))
    return out


def interpolate(sparse, dense, alpha: float) -> np.ndarray:
    sparse, dense = _f32(sparse), _f32(dense)
    out = np.empty_like(sparse)
    lib().ffo_interpolate(_p(sparse, C.c_float), _p(dense, C.c_float), sparse.shape[0], float(alpha), _p(out, C.c_float))
    return out


class OracleNotFound(IndexError):
    def __init__(self, pair: int):
        super().__init__(f"pair {pair} names an ordinal with no rows")
        self.pair = pair


def score(vectors, doc_off, doc_rows, psg_row, queries, pair_off, pair_ids, mode: int,
          sparse=None, alpha: float = 0.0):
    """Full CSR path (see ``ffo_score``).  Returns ``(ff, interp_or_None)``."""
    vectors, queries = _f32(vectors), _f32(queries)
    doc_off = np.ascontiguousarray(doc_off, dtype=np.int64)
    doc_rows = np.ascontiguousarray(doc_rows, dtype=np.int32)
    psg_row = np.ascontiguousarray(psg_row, dtype=np.int32)
    pair_off = np.ascontiguousarray(pair_off, dtype=np.int64)
    pair_ids = np.ascontiguousarray(pair_ids, dtype=np.int32)
    n_pairs = int(pair_off[-1])
    ff = np.empty(n_pairs, dtype=np.float32)
    sp = None if sparse is None else _f32(sparse)
    interp = None if sparse is None else np.empty(n_pairs, dtype=np.float32)
    bad = C.c_int64(-1)
    rc = lib().ffo_score(
        _p(vectors, C.c_float), vectors.shape[0], vectors.shape[1],
        _p(doc_off, C.c_int64), _p(doc_rows, C.c_int32), doc_off.shape[0] - 1,
        _p(psg_row, C.c_int32), psg_row.shape[0],
        _p(queries, C.c_float), queries.shape[0], _p(pair_off, C.c_int64), _p(pair_ids, C.c_int32),
        mode, _p(sp, C.c_float), float(alpha), _p(ff, C.c_float), _p(interp, C.c_float), C.byref(bad),
    )
    if rc == NOT_FOUND:
        raise OracleNotFound(bad.value)
    if rc != OK:
        raise ValueError(f"ffo_score failed with {rc}")
    return ff, interp


def order(scores, pair_off) -> np.ndarray:
    scores = _f32(scores)
    pair_off = np.ascontiguousarray(pair_off, dtype=np.int64)
    out = np.empty(scores.shape[0], dtype=np.int64)
    rc = lib().ffo_order(_p(scores, C.c_float), pair_off.shape[0] - 1, _p(pair_off, C.c_int64), _p(out, C.c_int64))
    if rc != OK:
        raise ValueError("ffo_order failed")
    return out


# --------------------------------------------------------------------------------------
# NumPy port: the reference's own three array operations (index/base.py:301-303) on
# integer CSR inputs.  This is what bench.py times as the reference-equivalent CPU path.
# --------------------------------------------------------------------------------------
def expand_pairs(doc_off, doc_rows, psg_row, pair_off, pair_ids, mode: int):
    """Pairs -> gathered rows (index/util.py:29-41).  Returns (row_ids, row_q, seg_off)."""
    doc_off = np.asarray(doc_off, dtype=np.int64)
    pair_off = np.asarray(pair_off, dtype=np.int64)
    pair_ids = np.asarray(pair_ids, dtype=np.int64)
    n_pairs = int(pair_off[-1])
    pair_q = np.repeat(np.arange(pair_off.shape[0] - 1, dtype=np.int32), np.diff(pair_off))
    if mode == MODE_PASSAGE:
        rows = np.asarray(psg_row, dtype=np.int32)[pair_ids]
        return rows, pair_q, np.arange(n_pairs + 1, dtype=np.int64)
    start = doc_off[pair_ids]
    if mode == MODE_FIRSTP:
        rows = np.asarray(doc_rows, dtype=np.int32)[start]
        return rows, pair_q, np.arange(n_pairs + 1, dtype=np.int64)
    cnt = doc_off[pair_ids + 1] - start
    seg_off = np.zeros(n_pairs + 1, dtype=np.int64)
    np.cumsum(cnt, out=seg_off[1:])
    within = np.arange(int(seg_off[-1]), dtype=np.int64) - np.repeat(seg_off[:-1], cnt)
    rows = np.asarray(doc_rows, dtype=np.int32)[np.repeat(start, cnt) + within]
    return rows, np.repeat(pair_q, cnt), seg_off


def numpy_port_row_scores(queries, vectors, row_q, row_ids) -> np.ndarray:
    """index/base.py:301-303 verbatim in effect: two fancy-index gathers, one product-sum."""
    q_reps = queries[row_q]
    d_reps = vectors[row_ids]
    return np.sum(q_reps * d_reps, axis=1)


def numpy_port_score(vectors, doc_off, doc_rows, psg_row, queries, pair_off, pair_ids, mode: int,
                     sparse=None, alpha: float = 0.0, batch_queries: int = 64):
    """Reference-equivalent CPU path on CSR inputs, batched over queries like
    ``Index.__call__(batch_size=...)`` (index/base.py:445-459)."""
    pair_off = np.asarray(pair_off, dtype=np.int64)
    n_q = pair_off.shape[0] - 1
    ff = np.empty(int(pair_off[-1]), dtype=np.float32)
    for q0 in range(0, n_q, batch_queries):
        q1 = min(n_q, q0 + batch_queries)
        lo, hi = int(pair_off[q0]), int(pair_off[q1])
        sub_off = pair_off[q0 : q1 + 1] - lo
        rows, row_q, seg_off = expand_pairs(doc_off, doc_rows, psg_row, sub_off, pair_ids[lo:hi], mode)
        rs = numpy_port_row_scores(queries[q0:q1], vectors, row_q, rows)
        if mode == MODE_MAXP:
            ff[lo:hi] = np.maximum.reduceat(rs, seg_off[:-1])
        elif mode == MODE_AVEP:
            ff[lo:hi] = aggregate(rs, seg_off, MODE_AVEP)
        else:
            ff[lo:hi] = rs[seg_off[:-1]]
    interp = None
    if sparse is not None:
        # ranking.py:319 with a Python-float alpha on float32 columns
        interp = alpha * np.asarray(sparse, dtype=np.float32) + (1 - alpha) * ff
    return ff, interp


# --------------------------------------------------------------------------------------
# The path as a user of the reference experiences it: frames of STRING ids, a Python look-up per distinct id,
# pandas merges and a group-by around the three NumPy lines, an outer merge for the interpolation and a sort of
# strings + floats for every Ranking made ("a pandas-faithful variant").  bench.py times it
# beside ``numpy_port_score`` on the same sample; tests check that it gives the same scores.
# --------------------------------------------------------------------------------------
def pandas_port_score(vectors, doc_off, doc_rows, psg_row, queries, pair_off, pair_ids, mode: int, sparse, alpha: float,
                      batch_queries: int | None = None):
    """``index(first)`` then ``first.interpolate(dense, alpha)`` through pandas, restating (not importing) what
    ``Index.__call__`` (index/base.py:389-469), ``_compute_scores`` (:279-314), ``get_indices`` (index/util.py:12-42),
    ``Ranking.__init__`` (ranking.py:67-121) and ``Ranking.interpolate`` (:293-326) do, operation by operation.
    Candidate ``c`` of the index is named ``"D<c>"`` (``"P<c>"`` in PASSAGE mode), query ``q`` ``"q<q>"``.

    Returns ``(ff, interp, seconds)``: dense and interpolated scores per candidate in the CSR order of the input, and a
    dict of the time spent per step.  The frames of strings and the id dictionaries are built before the clock starts
    (the reference holds them already: the run is a ``Ranking``, the dictionaries are the index)."""
    import time

    import pandas as pd

    pair_off = np.asarray(pair_off, dtype=np.int64)
    pair_ids = np.asarray(pair_ids, dtype=np.int64)
    doc_off = np.asarray(doc_off, dtype=np.int64)
    doc_rows = np.asarray(doc_rows, dtype=np.int64)
    psg_row = np.asarray(psg_row, dtype=np.int64)
    n_q = pair_off.shape[0] - 1
    lens = np.diff(pair_off)
    prefix = "P" if mode == MODE_PASSAGE else "D"
    # ---- what exists before the call (untimed) -----------------------------------------------------------------------
    touched = np.unique(pair_ids)
    if mode == MODE_PASSAGE:
        psg_to_row = {f"P{c}": int(psg_row[c]) for c in touched}
        doc_to_rows: dict = {}
    else:
        doc_to_rows = {f"D{c}": [int(r) for r in doc_rows[doc_off[c]:doc_off[c + 1]]] for c in touched}
        psg_to_row = {}
    q_names = np.array([f"q{q:07d}" for q in range(n_q)], dtype=object)
    first = pd.DataFrame({"q_id": np.repeat(q_names, lens), "id": np.array([f"{prefix}{c}" for c in pair_ids], dtype=object),
                          "score": np.asarray(sparse, dtype=np.float32), "query": np.repeat(q_names, lens)})
    position = pd.Series(np.arange(len(first)), index=pd.MultiIndex.from_arrays([first["q_id"], first["id"]]))
    first = first.sort_values(by=["q_id", "score"], ascending=False).reset_index(drop=True)  # a Ranking holds its frame sorted
    sec = {}
    clock = time.perf_counter

    def get_indices(ids):  # index/util.py:12-42, one dictionary look-up (or two) per distinct id
        rows, names = [], []
        for name in ids:
            if mode == MODE_PASSAGE:
                found = [psg_to_row[name]] if name in psg_to_row else []
            else:
                found = doc_to_rows.get(name, [])
                if mode == MODE_FIRSTP:
                    found = found[:1]
            if not found:
                raise IndexError(f"ID {name} not found in the index.")
            rows.extend(found)
            names.extend([name] * len(found))
        return rows, names

    def compute_scores(data, query_vectors):  # index/base.py:279-314
        rows, names = get_indices(data["id"].unique())
        got = vectors[rows]
        by_id = pd.DataFrame(names, columns=["id"])
        by_id["id_idx"] = by_id.index
        merged = by_id.merge(data[["id", "q_no"]], how="left", on="id")
        q_reps = query_vectors[merged["q_no"].tolist()]
        d_reps = got[merged["id_idx"].tolist()]
        merged["ff_score"] = np.sum(q_reps * d_reps, axis=1)
        op = "max" if mode == MODE_MAXP else "mean" if mode == MODE_AVEP else "first"
        return data.merge(merged.groupby(["id", "q_no"], as_index=False).aggregate(op), on=["id", "q_no"])

    # ---- Index.__call__ (index/base.py:418-469) ------------------------------------------------------------------------
    t0 = clock()
    numbered = first[["q_id", "query"]].drop_duplicates().reset_index(drop=True)
    numbered["q_no"] = numbered.index
    frame = first.merge(numbered, on=["q_id", "query"], suffixes=[None, "_"])
    q_vectors = np.asarray(queries, dtype=np.float32)[[int(name[1:]) for name in numbered["query"]]]  # (the encoder: a table look-up)
    sec["queries_numbered_s"] = clock() - t0
    t0 = clock()
    if batch_queries is None:
        result = compute_scores(frame, q_vectors)
    else:
        parts = [compute_scores(frame.loc[frame["q_no"].isin(range(a, a + batch_queries))], q_vectors) for a in range(0, len(numbered), batch_queries)]
        result = pd.concat(parts)
    sec["compute_scores_s"] = clock() - t0
    t0 = clock()
    result["score"] = result["ff_score"]
    dense = result[["q_id", "id", "score", "query"]].dropna().astype({"score": np.float32})
    dense = dense.sort_values(by=["q_id", "score"], ascending=False).reset_index(drop=True)  # Ranking.__init__ (ranking.py:103-117)
    sec["dense_ranking_made_s"] = clock() - t0
    # ---- Ranking.interpolate (ranking.py:293-326) ----------------------------------------------------------------------
    t0 = clock()
    both = first.merge(dense, on=["q_id", "id"], suffixes=(None, "_other"), how="outer").fillna(0)
    both["score"] = alpha * both["score"] + (1 - alpha) * both["score_other"]
    final = both[["q_id", "id", "score", "query"]].astype({"score": np.float32})
    final = final.sort_values(by=["q_id", "score"], ascending=False).reset_index(drop=True)
    sec["interpolated_ranking_made_s"] = clock() - t0
    sec["total_s"] = sum(sec.values())
    # ---- back to CSR order, for the checks (untimed) --------------------------------------------------------------------
    n = len(first)
    ff = np.empty(n, dtype=np.float32)
    interp = np.empty(n, dtype=np.float32)
    ff[position.loc[pd.MultiIndex.from_arrays([dense["q_id"], dense["id"]])].to_numpy()] = dense["score"].to_numpy()
    interp[position.loc[pd.MultiIndex.from_arrays([final["q_id"], final["id"]])].to_numpy()] = final["score"].to_numpy()
    return ff, interp, sec


# --------------------------------------------------------------------------------------
# Early stopping (index/base.py:316-387), stated per query with plain loops (small cases).
# --------------------------------------------------------------------------------------
def early_stopping_mask(dense, sparse, pair_off, cutoff: int, alpha: float, depths, promotion: str = "numpy1") -> np.ndarray:
    """Which candidates ``Index._early_stopping`` scores (and therefore keeps): a bool per pair.

    ``dense``: the dense score every candidate WOULD get (a candidate's score does not depend on early
    stopping); candidates of a query stand in first-stage rank order.  Depth intervals ``[a, b)`` over the
    sorted depths ``>= cutoff`` (base.py:339-342); after the first interval a query continues only while

        int_score.nlargest(cutoff).iat[-1] < alpha * score.iat[-1] + (1 - alpha) * ff_score.max()    (base.py:353-355)

    over the rows scored so far; no rows in an interval (every query stopped, or a repeated depth) ends the
    loop (base.py:368-369).  The right-hand side multiplies a Python float by np.float32 SCALARS, so its
    arithmetic is NumPy's scalar promotion: ``promotion="numpy1"`` (numpy < 2, the reference's pinned range):
    float64 products and sum; ``"numpy2"`` (NEP 50): float32 throughout.  ``int_score`` is a Series
    expression (base.py:377-379): float32 under either.  Pinned by tests/golden/early_stopping_np1.npz and
    early_stopping_np2.npz, which the reference produced under numpy 1.26.4 and 2.2.6."""
    if promotion not in ("numpy1", "numpy2"):
        raise ValueError(promotion)
    dense, sparse = _f32(dense), _f32(sparse)
    pair_off = np.asarray(pair_off, dtype=np.int64)
    n_q = pair_off.shape[0] - 1
    interp = interpolate(sparse, dense, alpha)
    scored = np.zeros(dense.shape[0], dtype=bool)
    active = [True] * n_q
    a = 0
    for b in sorted(int(d) for d in depths):
        if b < cutoff:
            continue
        if a > 0:
            for q in range(n_q):
                lo = int(pair_off[q])
                n = min(int(pair_off[q + 1]) - lo, a)
                if not active[q] or n == 0:
                    active[q] = False  # (a query without scored rows never reaches the reference's filter)
                    continue
                kth = np.sort(interp[lo:lo + n])[::-1][min(cutoff, n) - 1]
                last, best = sparse[lo + n - 1], dense[lo:lo + n].max()
                if promotion == "numpy1":
                    keep = float(kth) < float(alpha) * float(last) + float(1 - alpha) * float(best)  # Python floats are doubles
                else:
                    keep = kth < np.float32(np.float32(alpha) * last) + np.float32(np.float32(1 - alpha) * best)
                active[q] = bool(keep)
        took = False
        for q in range(n_q):
            lo, n = int(pair_off[q]), int(pair_off[q + 1] - pair_off[q])
            if active[q] and min(n, b) > min(n, a):
                scored[lo + min(n, a):lo + min(n, b)] = True
                took = True
        if not took:
            break
        a = b
    return scored
