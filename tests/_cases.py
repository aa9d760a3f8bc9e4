"""Rebuild the inputs of a golden fixture as integer CSR arrays (shared by CPU and GPU tests)."""

from __future__ import annotations

import hashlib
import json
from dataclasses import dataclass
from pathlib import Path

import numpy as np

from fast_forward_amd import synth

GOLDEN = Path(__file__).resolve().parent / "golden"
MODE_NO = {"PASSAGE": 1, "MAXP": 2, "FIRSTP": 3, "AVEP": 4}
ALPHAS = (0.05, 0.1, 0.2, 0.5)


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def manifest() -> dict:
    return json.loads((GOLDEN / "manifest.json").read_text())


@dataclass
class ScoringCase:
    name: str
    z: "np.lib.npyio.NpzFile"
    vectors: np.ndarray
    queries: np.ndarray
    doc_off: np.ndarray
    doc_rows: np.ndarray
    psg_row: np.ndarray
    doc_names: list
    psg_names: list
    q_names: list

    def pairs(self, mode: str):
        """(pair_off, pair_ids, sparse, id_names) with queries in fixture order (q_no = i)."""
        if mode == "PASSAGE":
            cands, sparse, prefix = self.z["cand_psgs"], self.z["sparse_psg"], "P"
        else:
            cands, sparse, prefix = self.z["cand_docs"], self.z["sparse_doc"], "D"
        n_q, k = cands.shape
        pair_off = np.arange(n_q + 1, dtype=np.int64) * k
        ids = cands.reshape(-1).astype(np.int32)
        names = [f"{prefix}{c}" for c in ids]
        return pair_off, ids, sparse.reshape(-1).astype(np.float32), names

    def modes(self):
        return [m for m in MODE_NO if f"{m}_ff_score" in self.z.files]

    def expected(self, key: str) -> dict:
        """{(q_id, id): fp32 score} of a stored reference frame."""
        z = self.z
        return {(q, i): s for q, i, s in zip(z[f"{key}_q_id"].tolist(), z[f"{key}_id"].tolist(), z[f"{key}_score"])}

    def expected_order(self, key: str) -> dict:
        """{q_id: [(id, score), ...]} in the reference's output order."""
        z = self.z
        out: dict = {}
        for q, i, s in zip(z[f"{key}_q_id"].tolist(), z[f"{key}_id"].tolist(), z[f"{key}_score"]):
            out.setdefault(q, []).append((i, s))
        return out


def load_scoring_case(name: str) -> ScoringCase:
    z = np.load(GOLDEN / f"{name}.npz")
    seed, dim, scale, n_rows = int(z["seed"]), int(z["dim"]), float(z["scale"]), int(z["n_rows"])
    n_q = z["cand_docs"].shape[0]
    store = np.dtype(str(z["store"])) if "store" in z.files else np.dtype(np.float32)
    vectors = synth.rows(seed, scale, 0, n_rows, dim).astype(store)  # dtype handed to Index.add
    queries = synth.rows(seed + 1000, scale, 0, n_q, dim)
    assert sha(vectors) == str(z["vectors_sha"]), "synthetic generator drifted from the fixture"
    assert sha(queries) == str(z["queries_sha"])
    doc_of_row = z["doc_of_row"]
    n_docs = int(doc_of_row.max()) + 1
    counts = np.bincount(doc_of_row, minlength=n_docs)
    doc_off = np.zeros(n_docs + 1, dtype=np.int64)
    np.cumsum(counts, out=doc_off[1:])
    doc_rows = np.argsort(doc_of_row, kind="stable").astype(np.int32)
    return ScoringCase(
        name=name, z=z, vectors=vectors, queries=queries, doc_off=doc_off, doc_rows=doc_rows,
        psg_row=np.arange(n_rows, dtype=np.int32),
        doc_names=[f"D{d}" for d in doc_of_row], psg_names=[f"P{r}" for r in range(n_rows)],
        q_names=z["q_names"].tolist(),
    )


SCORING_CASES = ["config1", "d768", "s2_sub", "dim5", "dim20", "dim100", "dim300", "dim384", "dim1024",
                 "d768_f16", "dim100_f16", "dim128_f16"]  # *_f16: float16 rows handed to the reference


def bits(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_same_order_modulo_ties(got_ids, got_scores, exp):
    """Both lists are sorted by score descending; they may differ only inside runs of
    exactly equal scores (ranking.py:116 uses an unstable sort)."""
    exp_ids = [i for i, _ in exp]
    exp_scores = np.array([s for _, s in exp], dtype=np.float32)
    got_scores = np.asarray(got_scores, dtype=np.float32)
    assert np.array_equal(bits(got_scores), bits(exp_scores)), "score sequences differ"
    start = 0
    n = len(exp_ids)
    while start < n:
        end = start + 1
        while end < n and exp_scores[end] == exp_scores[start]:
            end += 1
        assert sorted(got_ids[start:end]) == sorted(exp_ids[start:end])
        start = end


# ---- the reference's DEFAULT arithmetic (float64 chunks, index/memory.py:106) ------------------------
F64_CASES = ["multichunk_f64", "multichunk_f64_unit", "multichunk_f64_d128", "multichunk_f64_d300"]
F64_TOL = 1e-4      # north_star: scores match the reference's to 1e-4 (fp32)
F64_ALPHAS = (0.1, 0.5)


def tolerance_report(got: np.ndarray, want: np.ndarray, pair_off: np.ndarray) -> dict:
    """How fp32 scores `got` sit against the float64-accumulated reference scores `want`
    (both float32 arrays over the same pairs): the largest absolute difference, the bar it is held
    to, and how the two orderings differ -- the number of adjacent inversions (pairs that the two
    sides order differently although their reference scores differ) and the widest reference gap
    any of them spans."""
    got, want = np.asarray(got, dtype=np.float32), np.asarray(want, dtype=np.float32)
    scale = max(1.0, float(np.abs(want).max())) if want.size else 1.0
    inversions, widest = 0, 0.0
    for q in range(len(pair_off) - 1):
        lo, hi = int(pair_off[q]), int(pair_off[q + 1])
        g, w = got[lo:hi], want[lo:hi]
        mine = np.argsort(-g.astype(np.float64), kind="stable")
        w_in_my_order = w[mine].astype(np.float64)
        bad = np.flatnonzero(w_in_my_order[:-1] < w_in_my_order[1:])  # the reference puts these two the other way round
        inversions += int(bad.size)
        if bad.size:
            widest = max(widest, float((w_in_my_order[bad + 1] - w_in_my_order[bad]).max()))
    return {"max_abs_diff": float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max()) if want.size else 0.0,
            "bar": F64_TOL * scale, "order_inversions": inversions, "widest_inverted_gap": widest, "pairs": int(want.size)}


def assert_within_reference_tolerance(got, want, pair_off, what: str) -> dict:
    """The parity bar against the reference's float64 path: |diff| <= 1e-4 * max(1, |score|_inf), and
    the orderings agree except inside reference score gaps below 2e-4 * that scale."""
    rep = tolerance_report(got, want, pair_off)
    assert rep["max_abs_diff"] <= rep["bar"], f"{what}: {rep}"
    assert rep["widest_inverted_gap"] < 2 * rep["bar"], f"{what}: {rep}"
    return rep


def load_sub200k() -> dict:
    """The sub-sampled config 2 (oracle/gen_golden.py::sub200k_case): 200 k rows x
    768, Q = 200, k = 1000, MaxP, scored by the reference as one fp32 chunk and with its default
    (float64) chunking.  Inputs come back from the seeds; vectors are NOT materialised here (614 MB:
    ``synth.rows(seed, scale, 0, n_rows, dim)`` on the host, ``fill_synthetic`` on the device)."""
    z = np.load(GOLDEN / "sub200k.npz")
    seed, dim, scale, n_rows, n_docs, n_q, k = (int(z["seed"]), int(z["dim"]), float(z["scale"]), int(z["n_rows"]), int(z["n_docs"]),
                                                int(z["n_q"]), int(z["k"]))
    per = (synth.uniform_u32(np.arange(200_000, dtype=np.uint64), seed + 7) % np.uint32(8)).astype(np.int64) + 1
    per = per[:n_docs]
    assert int(per.sum()) == n_rows
    doc_off = np.concatenate([[0], np.cumsum(per)]).astype(np.int64)
    stratum = n_docs // k
    idx_u = np.arange(n_q * k, dtype=np.uint64)
    cand = ((np.arange(n_q * k) % k) * stratum + (synth.uniform_u32(idx_u, seed + 1) % np.uint32(stratum)).astype(np.int64)).astype(np.int32)
    sparse = (synth.uniform_u32(idx_u, seed + 2).astype(np.float32) * np.float32(25.0 / 2**32)).astype(np.float32)
    assert sha(cand.reshape(n_q, k)) == str(z["cand_sha"]) and sha(sparse.reshape(n_q, k)) == str(z["sparse_sha"])
    queries = synth.rows(seed + 1000, scale, 0, n_q, dim)
    assert sha(queries) == str(z["queries_sha"])
    return {"seed": seed, "dim": dim, "scale": scale, "n_rows": n_rows, "n_docs": n_docs, "doc_off": doc_off,
            "pair_off": np.arange(n_q + 1, dtype=np.int64) * k, "pair_ids": cand, "sparse": sparse, "queries": queries,
            "vectors_sha": str(z["vectors_sha"]), "ff_f32chunk": z["ff_f32chunk"].reshape(-1), "ff_default_chunks": z["ff_default_chunks"].reshape(-1)}


# ---- product-quantized fixtures (oracle/gen_golden.py::pq_case) ------------------------------------
PQ_CASES = ["pq_d768_m8", "pq_d768_m16", "pq_d1024_m32", "pq_d60_m5"]


@dataclass
class PQCase(ScoringCase):
    codes: np.ndarray = None       # uint8 [n_rows, M], as the reference's index stored them
    codewords: np.ndarray = None   # float32 [M, Ks, Ds]
    raw_vectors: np.ndarray = None  # what was handed to Index.add (before encoding)


def load_pq_case(name: str) -> PQCase:
    """``vectors`` of the returned case are the DECODED rows (what the reference scores)."""
    z = np.load(GOLDEN / f"{name}.npz")
    seed, dim, n_rows, M, Ks = int(z["seed"]), int(z["dim"]), int(z["n_rows"]), int(z["M"]), int(z["Ks"])
    Ds = dim // M
    codewords = synth.rows(seed + 500, 1.0, 0, M * Ks, Ds).reshape(M, Ks, Ds)
    assert sha(codewords) == str(z["codewords_sha"])
    raw = synth.rows(seed, 1.0, 0, n_rows, dim)
    queries = synth.rows(seed + 1000, 1.0, 0, z["cand_docs"].shape[0], dim)
    assert sha(raw) == str(z["vectors_sha"]) and sha(queries) == str(z["queries_sha"])
    codes = z["codes"]
    decoded = np.concatenate([codewords[m][codes[:, m]] for m in range(M)], axis=1)
    doc_of_row = z["doc_of_row"]
    n_docs = int(doc_of_row.max()) + 1
    doc_off = np.zeros(n_docs + 1, dtype=np.int64)
    np.cumsum(np.bincount(doc_of_row, minlength=n_docs), out=doc_off[1:])
    return PQCase(
        name=name, z=z, vectors=decoded, queries=queries, doc_off=doc_off,
        doc_rows=np.argsort(doc_of_row, kind="stable").astype(np.int32), psg_row=np.arange(n_rows, dtype=np.int32),
        doc_names=[f"D{d}" for d in doc_of_row], psg_names=[f"P{r}" for r in range(n_rows)], q_names=z["q_names"].tolist(),
        codes=codes, codewords=codewords, raw_vectors=raw,
    )


def run_file_text(seed: int = 77) -> tuple[str, dict]:
    """A TREC run file big enough (>= 65 536 rows, >= 256 KB) for ``Ranking.from_file`` to read it natively into
    the coded form: 150 queries in shuffled order, 300-800 candidates each in shuffled order, distinct scores
    inside a query (the reference leaves the order of exact ties open), negative and tiny scores, some non-ASCII
    ids, integer-looking ids for some queries.  Returns (text, queries).  Used by oracle/gen_golden.py (fed to the
    reference) and by the test that compares the build with the stored result."""
    rng = np.random.default_rng(seed)
    lines = []
    q_ids = [f"q{i:03d}" for i in rng.permutation(150)]
    for q in q_ids:
        k = int(rng.integers(300, 800))
        docs = rng.choice(20_000, size=k, replace=False)
        scores = rng.permutation(k).astype(np.float64) * 0.0625 - 7.0  # distinct, exactly representable, both signs
        scores[0] = 1e-7 * (1 + int(q[1:]))
        for i, (d, s) in enumerate(zip(docs, scores)):
            name = f"dö{d}" if d % 13 == 0 else (f"{d}" if int(q[1:]) % 10 == 0 else f"D{d}")
            lines.append(f"{q} Q0 {name} {i + 1} {float(s)!r} bm25")
    return "\n".join(lines) + "\n", {q: f"text of {q}" for q in q_ids}


def two_runs(seed: int = 91):
    """Two first-stage runs as frames of (q_id, id, score) that are NOT each other's image -- partly overlapping
    candidate sets, a query only the first has, one only the second has, the second one shuffled -- with more than
    65 536 rows together, so that ``Ranking.interpolate`` / ``+`` take the threaded outer join.  Scores are multiples
    of 1/8 (exact in float32, many exact ties after combining).  Used by oracle/gen_golden.py (fed to the reference)
    and by the test that compares the build with the stored result."""
    import pandas as pd

    rng = np.random.default_rng(seed)
    rows_a, rows_b = [], []
    for i in range(110):
        q = f"q{i:03d}"
        pool = rng.choice(25_000, size=650, replace=False)
        if i != 7:
            rows_a += [(q, f"D{d}", float(rng.integers(0, 160)) / 8.0) for d in pool[:450]]
        if i != 11:
            rows_b += [(q, f"D{d}", float(rng.integers(-24, 24)) / 8.0) for d in pool[200:]]
    order = rng.permutation(len(rows_b))
    rows_b = [rows_b[j] for j in order]
    fa = pd.DataFrame(rows_a, columns=["q_id", "id", "score"])
    fb = pd.DataFrame(rows_b, columns=["q_id", "id", "score"])
    return fa, fb


# ---- early stopping under both NumPy scalar-promotion regimes (oracle/gen_golden_promotion.py) ------------------
ES_PROMOTIONS = {"numpy1": "early_stopping_np1.npz", "numpy2": "early_stopping_np2.npz"}
ES_PARTS = ("small", "large", "docs")


def load_es_promotion_part(promotion: str, part: str) -> dict:
    """Inputs (from the early_stopping*.npz fixtures) and, per configuration, the rows the REFERENCE kept under the
    given NumPy (``{(q_id, id): score}``).  ``docs``: document runs with a mode per configuration."""
    import json

    from fast_forward_amd import synth

    out_file = np.load(GOLDEN / ES_PROMOTIONS[promotion])
    assert str(out_file["numpy"]).startswith("1." if promotion == "numpy1" else "2.")
    z = np.load(GOLDEN / {"small": "early_stopping.npz", "large": "early_stopping_large.npz", "docs": "early_stopping_docs.npz"}[part])
    dim, seed, scale = int(z["dim"]), int(z["seed"]), float(z["scale"])
    q_names, lens = z["q_names"].tolist(), z["lens"].tolist()
    n_rows = int(z["n_rows"] if part == "docs" else z["n_psg"])
    configs = [tuple(c) if part == "docs" else ("PASSAGE", *c) for c in json.loads(str(out_file[f"{part}_configs"]))]
    expected = [{(q, i): s for q, i, s in zip(out_file[f"{part}_es{ci}_q_id"].tolist(), out_file[f"{part}_es{ci}_id"].tolist(), out_file[f"{part}_es{ci}_score"])}
                for ci in range(len(configs))]
    return {"q_names": q_names, "lens": lens, "cands": z["cands"], "sparse": z["sparse"], "prefix": "D" if part == "docs" else "P",
            "vectors": synth.rows(seed, scale, 0, n_rows, dim), "queries": synth.rows(seed + 1000, scale, 0, len(q_names), dim),
            "doc_of_row": z["doc_of_row"] if part == "docs" else None, "n_rows": n_rows, "configs": configs, "expected": expected}
