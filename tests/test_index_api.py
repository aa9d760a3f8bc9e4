"""``InMemoryIndex`` / ``Index.__call__`` through the public API on the GPU, against what the
reference's own tests pin (reference tests/test_index.py:104-349, :376-387) and against frames
the reference produced (tests/golden).  Run with ``-m gpu``."""

import itertools
import json

import numpy as np
import pandas as pd
import pytest

from _cases import ALPHAS, GOLDEN, MODE_NO, assert_same_order_modulo_ties, bits, load_scoring_case
from fast_forward_amd import Ranking, synth
from fast_forward_amd.encoder import LambdaEncoder
from fast_forward_amd.index import InMemoryIndex, Mode

pytestmark = pytest.mark.gpu

QUERIES = {"q1": "query 1", "q2": "query 2"}
DOC_IDS = ["d0", "d0", "d1", "d2", "d3"]
PSG_IDS = ["p0", "p1", "p2", "p3", "p4"]
VECTORS = np.tril(np.ones((5, 5), dtype=np.int64))  # rows 1,0,0,0,0 / 1,1,0,0,0 / ...
DOC_RUN = {"q1": {"d0": 100, "d1": 2, "d2": 3, "d3": 200}, "q2": {"d0": 400, "d1": 5, "d2": 6, "d3": 800}}
PSG_RUN = {"q1": {"p0": 100, "p1": 2, "p2": 3, "p3": 4, "p4": 5}, "q2": {"p0": 500, "p1": 6, "p2": 7, "p3": 8, "p4": 9}}
ONES = LambdaEncoder(lambda _: np.array([1, 1, 1, 1, 1]))


def both_ids_index(**kw):
    idx = InMemoryIndex(ONES, **kw)
    idx.add(VECTORS, doc_ids=DOC_IDS, psg_ids=PSG_IDS)
    return idx


def partial_ids_index(**kw):
    """Some vectors have only a doc id, some only a psg id, some both (reference test_index.py:59-69)."""
    idx = InMemoryIndex(ONES, **kw)
    idx.add(VECTORS, doc_ids=[None, None] + DOC_IDS[2:], psg_ids=PSG_IDS[:-2] + [None, None])
    idx.add(VECTORS[:2], doc_ids=DOC_IDS[:2])
    idx.add(VECTORS[-2:], psg_ids=PSG_IDS[-2:])
    return idx


def test_properties():
    idx = both_ids_index()
    assert idx.doc_ids == set(DOC_IDS) and idx.psg_ids == set(PSG_IDS) and len(idx) == 5 and idx.dim == 5
    part = partial_ids_index()
    assert part.doc_ids == set(DOC_IDS) and part.psg_ids == set(PSG_IDS) and len(part) == 9 and part.dim == 5
    docs_only = InMemoryIndex()
    docs_only.add(VECTORS, doc_ids=DOC_IDS)
    assert docs_only.doc_ids == set(DOC_IDS) and len(docs_only.psg_ids) == 0


def test_known_answers_all_modes():
    idx, part = both_ids_index(), partial_ids_index()
    doc_ranking = Ranking.from_run(DOC_RUN, queries=QUERIES)
    psg_ranking = Ranking.from_run(PSG_RUN, queries=QUERIES)
    idx.mode = Mode.MAXP
    out = idx(doc_ranking)
    assert out.has_queries and out.name == "fast-forward"
    assert out == Ranking.from_run({q: {"d0": 2, "d1": 3, "d2": 4, "d3": 5} for q in QUERIES})
    idx.mode = Mode.FIRSTP
    assert idx(doc_ranking) == Ranking.from_run({q: {"d0": 1, "d1": 3, "d2": 4, "d3": 5} for q in QUERIES})
    avep = Ranking.from_run({q: {"d0": 1.5, "d1": 3, "d2": 4, "d3": 5} for q in QUERIES})
    idx.mode = part.mode = Mode.AVEP
    assert idx(doc_ranking) == avep and part(doc_ranking) == avep
    passage = Ranking.from_run({q: {f"p{i}": i + 1 for i in range(5)} for q in QUERIES})
    idx.mode = part.mode = Mode.PASSAGE
    assert idx(psg_ranking) == passage and part(psg_ranking) == passage


def test_error_contract():
    idx = both_ids_index()
    empty = InMemoryIndex()
    with pytest.raises(RuntimeError):  # duplicate passage id
        empty.add(VECTORS[:1], psg_ids=PSG_IDS[:1])
        empty.add(VECTORS[:1], psg_ids=PSG_IDS[:1])
    assert len(empty) == 1
    two_d = InMemoryIndex()
    two_d.add(np.array([[0, 0], [1, 1]]), doc_ids=["d1", "d2"])
    with pytest.raises(ValueError):  # wrong dimensionality
        two_d.add(np.array([[0, 0, 0], [1, 1, 1]]), doc_ids=["d3", "d4"])
    with pytest.raises(ValueError):
        idx(Ranking.from_run(DOC_RUN))  # no queries
    with pytest.raises(IndexError, match="ID dx not found in the index."):
        idx(Ranking.from_run({"q1": {"d0": 100, "dx": 2}}, queries=QUERIES))
    with pytest.raises(IndexError):
        idx._get_vectors(["nope"])
    from fast_forward_amd.quantizer import NanoPQ

    trained = NanoPQ(1, 2)
    trained.fit(np.random.default_rng(0).normal(size=(8, 5)).astype(np.float32), iter=1)
    with pytest.raises(RuntimeError):
        idx.quantizer = trained  # non-empty index
    with pytest.raises(RuntimeError):
        InMemoryIndex(quantizer=NanoPQ(1, 2))  # not fitted yet


def test_batch_size_invariance():
    idx = both_ids_index()
    r = Ranking.from_run(
        {"q1": {"d0": 2, "d1": 3, "d2": 4, "d3": 10}, "q2": {"d0": 5, "d1": 4, "d2": 3, "d3": 12},
         "q3": {"d0": 8, "d1": 5, "d2": 2, "d3": 1}, "q4": {"d0": 11, "d1": 6, "d2": 1, "d3": 2},
         "q5": {"d0": 14, "d1": 7, "d2": 0, "d3": 3}},
        queries={f"q{n}": f"query {n}" for n in range(1, 6)})
    expected = idx(r)
    for bs in (1, 2, 3, 5, 10):  # 1 and 5 divide Q: the reference raises there (base.py:455); here they work
        got = idx(r, batch_size=bs)
        assert got == expected
        # not only the same triples: the same frame, row for row (order, dtypes, query column, numbering)
        pd.testing.assert_frame_equal(got._df, expected._df)
        assert got._numbering()[1] == expected._numbering()[1] and np.array_equal(got._numbering()[0], expected._numbering()[0])


def test_add_retrieve_while_growing():
    idx = InMemoryIndex(init_size=8, alloc_size=4)
    data = np.random.default_rng(0).normal(size=(80, 16))
    doc_ids = [f"doc_{i // 2}" for i in range(80)]
    psg_ids = [f"psg_{i}" for i in range(80)]
    for lo, hi in [(0, 8), (8, 24), (24, 80)]:
        idx.add(data[lo:hi], doc_ids=doc_ids[lo:hi], psg_ids=psg_ids[lo:hi])
        assert len(idx) == hi
        idx.mode = Mode.PASSAGE
        vecs, ids = idx._get_vectors(psg_ids[lo:hi])
        assert ids == psg_ids[lo:hi] and np.allclose(vecs, data[lo:hi].astype(np.float32))
        idx.mode = Mode.MAXP
        vecs, ids = idx._get_vectors([f"doc_{i}" for i in range(lo // 2, hi // 2)])
        assert ids == doc_ids[lo:hi] and np.allclose(vecs, data[lo:hi].astype(np.float32))
    idx.consolidate()
    idx.mode = Mode.FIRSTP
    vecs, ids = idx._get_vectors(["doc_3"])
    assert ids == ["doc_3"] and np.allclose(vecs[0], data[6].astype(np.float32))


def test_iteration():
    idx = both_ids_index()
    for bs in (1, 3, 5, 10):
        vectors, doc_ids, psg_ids = zip(*idx.batch_iter(bs))
        assert np.array_equal(np.concatenate(vectors), VECTORS.astype(np.float32))
        assert list(itertools.chain.from_iterable(doc_ids)) == DOC_IDS
        assert list(itertools.chain.from_iterable(psg_ids)) == PSG_IDS
    assert [d for _, d, _ in idx] == DOC_IDS


def test_early_stopping_known_answer():
    """reference tests/test_index.py:273-333: only ranks 10..19 get scored."""
    idx = InMemoryIndex(LambdaEncoder(lambda q: np.array([10, 10])), mode=Mode.PASSAGE)
    idx.add(np.stack([[1, 0], [1, 1]] * 10), psg_ids=[f"p{i}" for i in range(20)])
    r = Ranking(pd.DataFrame([{"q_id": q, "query": q, "id": f"p{i}", "score": i} for i in range(20) for q in ("q1", "q2")]))
    expected = Ranking(pd.DataFrame(
        [{"q_id": q, "id": f"p{i}", "score": 20.0 if i % 2 else 10.0} for q in ("q1", "q2") for i in range(10, 20)]))
    for depths in ((2, 5, 10, 20), (5, 2, 20, 10)):
        got = idx(r, early_stopping=5, early_stopping_alpha=0.5, early_stopping_depths=depths)
        assert got == expected


def test_early_stopping_looks_up_only_what_it_scores():
    """As in the reference (get_indices runs per scored chunk, index/base.py:362-370), an id that is
    never reached is never looked up: an unknown id below the stopping depth is harmless, one
    inside a scored interval is an IndexError; batching gives the same result."""
    idx = InMemoryIndex(LambdaEncoder(lambda q: np.array([10, 10])), mode=Mode.PASSAGE)
    idx.add(np.stack([[1, 0], [1, 1]] * 10), psg_ids=[f"p{i}" for i in range(20)])
    rows = [{"q_id": q, "query": q, "id": f"p{i}", "score": i} for i in range(20) for q in ("q1", "q2")]
    # sparse scores 0..19: ranks 0..9 are p19..p10; the candidates with score < 10 are never reached
    deep = [dict(r, id="unknown-" + r["id"]) if r["score"] < 10 else r for r in rows]
    expected = Ranking(pd.DataFrame(
        [{"q_id": q, "id": f"p{i}", "score": 20.0 if i % 2 else 10.0} for q in ("q1", "q2") for i in range(10, 20)]))
    for batch_size in (None, 1):
        got = idx(Ranking(pd.DataFrame(deep)), early_stopping=5, early_stopping_alpha=0.5, early_stopping_depths=(5, 10, 20),
                  batch_size=batch_size)
        assert got == expected
        assert got._df["score"].is_monotonic_decreasing or len(got) == 2  # per-query order is checked below
        for q in ("q1", "q2"):
            assert got._df.loc[got._df["q_id"] == q, "score"].is_monotonic_decreasing
    shallow = [dict(r, id="unknown") if (r["score"] == 15 and r["q_id"] == "q2") else r for r in rows]
    with pytest.raises(IndexError, match="ID unknown not found in the index."):
        idx(Ranking(pd.DataFrame(shallow)), early_stopping=5, early_stopping_alpha=0.5, early_stopping_depths=(5, 10, 20))
    with pytest.raises(IndexError):  # without early stopping every id is needed
        idx(Ranking(pd.DataFrame(deep)))


@pytest.mark.parametrize("fixture", ["early_stopping", "early_stopping_large"])
def test_early_stopping_matches_reference_frames(fixture):
    """Frames the reference produced with early stopping on passage runs (a small run and one of 7730 rows, 24 queries of
    150 - 520 candidates): same rows, same bits, same order of the queries.  Every ranking takes the one route: integer id
    codes in, a coded ``Ranking`` out."""
    z = np.load(GOLDEN / f"{fixture}.npz")
    dim, seed, scale, n_psg = int(z["dim"]), int(z["seed"]), float(z["scale"]), int(z["n_psg"])
    q_names = z["q_names"].tolist()
    qvecs = synth.rows(seed + 1000, scale, 0, len(q_names), dim)
    idx = InMemoryIndex(LambdaEncoder(lambda t: qvecs[int(t)]), mode=Mode.PASSAGE)
    idx.early_stopping_promotion = "numpy2"  # these frames came out of the reference under numpy 2.2.6 (both regimes: next test but one)
    idx.add(synth.rows(seed, scale, 0, n_psg, dim), psg_ids=[f"P{r}" for r in range(n_psg)])
    rows, pos = [], 0
    for qi, n in enumerate(z["lens"].tolist()):
        for c, s in zip(z["cands"][pos:pos + n], z["sparse"][pos:pos + n]):
            rows.append((q_names[qi], f"P{c}", s))
        pos += n
    first = Ranking(pd.DataFrame(rows, columns=["q_id", "id", "score"]), queries={q: str(i) for i, q in enumerate(q_names)})
    for ci, (cutoff, alpha, depths) in enumerate(json.loads(str(z["configs"]))):
        res = idx(first, early_stopping=cutoff, early_stopping_alpha=alpha, early_stopping_depths=depths)
        assert res._coded is not None
        got = res._df
        assert got["q_id"].tolist() == z[f"es{ci}_q_id"].tolist()  # same query order as the reference's frame
        assert got["query"].tolist() == [str(q_names.index(q)) for q in got["q_id"]]
        want = {(q, i): s for q, i, s in zip(z[f"es{ci}_q_id"].tolist(), z[f"es{ci}_id"].tolist(), z[f"es{ci}_score"])}
        have = {(q, i): s for q, i, s in zip(got["q_id"], got["id"], got["score"])}
        assert have.keys() == want.keys(), f"config {ci}: scored sets differ"
        ks = sorted(want)
        assert np.array_equal(bits(np.array([have[k] for k in ks])), bits(np.array([want[k] for k in ks]))), ci


@pytest.mark.parametrize("criterion", ["device", "host"])
def test_early_stopping_on_document_runs_matches_reference_frames(criterion, monkeypatch):
    """Frames the reference produced with early stopping on DOCUMENT runs (MaxP / AveP / FirstP over 1..6
    passages per document; oracle/gen_golden.py::early_stopping_docs_case): same scored set, same bits --
    through the device-resident criterion (ffx_plan_early_stopping) and through its NumPy statement."""
    from fast_forward_amd.index.api import Index

    z = np.load(GOLDEN / "early_stopping_docs.npz")
    dim, seed, scale, n_rows = int(z["dim"]), int(z["seed"]), float(z["scale"]), int(z["n_rows"])
    q_names = z["q_names"].tolist()
    qvecs = synth.rows(seed + 1000, scale, 0, len(q_names), dim)
    vectors = synth.rows(seed, scale, 0, n_rows, dim)
    rows, pos = [], 0
    for qi, n in enumerate(z["lens"].tolist()):
        for c, s in zip(z["cands"][pos:pos + n], z["sparse"][pos:pos + n]):
            rows.append((q_names[qi], f"D{c}", s))
        pos += n
    first = Ranking(pd.DataFrame(rows, columns=["q_id", "id", "score"]), queries={q: str(i) for i, q in enumerate(q_names)})
    if criterion == "host":  # the NumPy statement of the criterion instead of the device-resident one
        monkeypatch.setattr(InMemoryIndex, "_early_stopping_kept", Index._early_stopping_kept)
    for ci, (mname, cutoff, alpha, depths) in enumerate(json.loads(str(z["configs"]))):
        idx = InMemoryIndex(LambdaEncoder(lambda t: qvecs[int(t)]), mode=Mode(MODE_NO[mname]))
        idx.early_stopping_promotion = "numpy2"  # (frames of the reference under numpy 2.2.6)
        idx.add(vectors, doc_ids=[f"D{d}" for d in z["doc_of_row"]])
        res = idx(first, early_stopping=cutoff, early_stopping_alpha=alpha, early_stopping_depths=depths)
        got = res._df
        want = {(q, i): s for q, i, s in zip(z[f"es{ci}_q_id"].tolist(), z[f"es{ci}_id"].tolist(), z[f"es{ci}_score"])}
        have = {(q, i): s for q, i, s in zip(got["q_id"], got["id"], got["score"])}
        assert have.keys() == want.keys(), f"config {ci} ({mname}): scored sets differ"
        ks = sorted(want)
        assert np.array_equal(bits(np.array([have[k] for k in ks])), bits(np.array([want[k] for k in ks]))), (ci, mname)
        idx.close()


@pytest.mark.parametrize("route", ["device", "host"])
@pytest.mark.parametrize("part", ["small", "large", "docs"])
@pytest.mark.parametrize("promotion", ["numpy1", "numpy2"])
def test_early_stopping_matches_the_reference_under_both_numpy_promotions(promotion, part, route, monkeypatch):
    """What the reference keeps depends on NumPy's scalar promotion (base.py:353-355 multiplies a Python float by a
    np.float32 scalar): float64 under its pinned numpy < 2 (the default here, ``early_stopping_promotion = "numpy1"``),
    float32 under numpy >= 2.  Both are pinned by frames the REFERENCE produced under numpy 1.26.4 and 2.2.6
    (oracle/gen_golden_promotion.py): ordinary configurations (one of which the two regimes already disagree on),
    repeated depths, and cut-off 1 after a first depth of 1, where numpy >= 2 never continues.  Same rows, same bits,
    through the device-resident criterion and through its NumPy statement."""
    from _cases import load_es_promotion_part
    from fast_forward_amd.index.api import Index

    c = load_es_promotion_part(promotion, part)
    if route == "host":
        monkeypatch.setattr(InMemoryIndex, "_early_stopping_kept", Index._early_stopping_kept)
    rows, pos = [], 0
    for qi, n in enumerate(c["lens"]):
        rows += [(c["q_names"][qi], f"{c['prefix']}{cand}", s) for cand, s in zip(c["cands"][pos:pos + n], c["sparse"][pos:pos + n])]
        pos += n
    first = Ranking(pd.DataFrame(rows, columns=["q_id", "id", "score"]), queries={q: str(i) for i, q in enumerate(c["q_names"])})
    qvecs = c["queries"]
    indexes = {}
    for (mname, cutoff, alpha, depths), want in zip(c["configs"], c["expected"]):
        if mname not in indexes:
            idx = indexes[mname] = InMemoryIndex(LambdaEncoder(lambda t: qvecs[int(t)]), mode=Mode(MODE_NO[mname]))
            idx.early_stopping_promotion = promotion
            if part == "docs":
                idx.add(c["vectors"], doc_ids=[f"D{d}" for d in c["doc_of_row"]])
            else:
                idx.add(c["vectors"], psg_ids=[f"P{r}" for r in range(c["n_rows"])])
        res = indexes[mname](first, early_stopping=cutoff, early_stopping_alpha=alpha, early_stopping_depths=depths)
        got = res._df
        have = {(q, i): s for q, i, s in zip(got["q_id"], got["id"], got["score"])}
        assert have.keys() == want.keys(), (promotion, part, route, mname, cutoff, alpha, depths, len(have), len(want))
        ks = sorted(want)
        assert np.array_equal(bits(np.array([have[k] for k in ks], dtype=np.float32)), bits(np.array([want[k] for k in ks], dtype=np.float32)))
    for idx in indexes.values():
        idx.close()


def test_early_stopping_promotion_must_be_one_of_the_two():
    idx = InMemoryIndex(LambdaEncoder(lambda q: np.array([10, 10])), mode=Mode.PASSAGE)
    idx.add(np.stack([[1, 0], [1, 1]] * 10), psg_ids=[f"p{i}" for i in range(20)])
    r = Ranking(pd.DataFrame([{"q_id": "q", "query": "q", "id": f"p{i}", "score": i} for i in range(20)]))
    assert InMemoryIndex.early_stopping_promotion == "numpy1"  # the reference's pinned range is the default
    idx.early_stopping_promotion = "numpy3"
    with pytest.raises(ValueError):
        idx(r, early_stopping=5, early_stopping_alpha=0.5, early_stopping_depths=(5, 10))
    idx.close()


@pytest.mark.parametrize("mode", ["MAXP", "PASSAGE", "AVEP"])
def test_device_resident_early_stopping_equals_the_host_criterion(mode, monkeypatch):
    """ffx_plan_early_stopping (per-query state in HBM between depth intervals, radix-select of the
    cutoff-th best score) against the NumPy statement of base.py:316-387 in Index._early_stopping_kept,
    which the golden frames pin: same scored set, same bits -- ragged queries, documents of 1..8
    passages, depths in any order, cutoffs from 1 to beyond the list length."""
    from fast_forward_amd.index.api import Index

    def host_statement(*args):  # the same call with the NumPy statement of the criterion
        with monkeypatch.context() as m:
            m.setattr(InMemoryIndex, "_early_stopping_kept", Index._early_stopping_kept)
            return idx._early_stopping_scores(*args)

    case = load_scoring_case("s2_sub")
    qvecs = case.queries
    idx = InMemoryIndex(LambdaEncoder(lambda t: qvecs[int(t)]), mode=Mode(MODE_NO[mode]))
    idx.add(case.vectors, doc_ids=case.doc_names, psg_ids=case.psg_names)
    pair_off, _, sparse, names = case.pairs(mode)
    rng = np.random.default_rng(4)
    n_q = len(case.q_names)
    keep = np.ones(len(names), dtype=bool)
    for q in range(n_q):  # ragged: 0 .. k candidates
        k = pair_off[q + 1] - pair_off[q]
        keep[pair_off[q] + int(rng.integers(0, k + 1)):pair_off[q + 1]] = False
    keep[pair_off[2]:pair_off[3]] = False  # one query without candidates at all
    ids = np.array(names, dtype=object)[keep]
    q_no = np.repeat(np.arange(n_q), np.diff(pair_off))[keep]
    sp = np.empty(keep.sum(), dtype=np.float32)
    for q in range(n_q):  # a first-stage run is sorted by score inside a query
        sel = q_no == q
        sp[sel] = -np.sort(-sparse[keep][sel])
    for promotion, (cutoff, alpha, depths) in itertools.product(("numpy1", "numpy2"), (
            (10, 0.2, (20, 50, 100, 400)), (1, 0.5, (1, 2, 3, 5, 8, 13, 400)), (25, 0.05, (100, 30, 1000)),
            (7, 0.9, (400, 8)), (500, 0.3, (600, 700)), (3, 0.0, (5, 50)), (3, 1.0, (5, 50)), (2, 0.4, (2, 6, 6, 90)))):
        idx.early_stopping_promotion = promotion
        got = idx._early_stopping_scores(ids, q_no, sp, qvecs, cutoff, alpha, depths)
        stats = dict(idx.last_stats)
        assert stats["early_stopping_intervals"] >= 1
        want = host_statement(ids, q_no, sp, qvecs, cutoff, alpha, depths)
        assert np.array_equal(np.isnan(got), np.isnan(want)), (cutoff, alpha, depths)
        assert np.array_equal(bits(got[~np.isnan(got)]), bits(want[~np.isnan(want)])), (cutoff, alpha, depths)
        assert stats["early_stopping_scored_pairs"] == int((~np.isnan(want)).sum())
    # rows of a query interleaved with other queries' rows (frame not grouped): same scores per row
    order = np.argsort(rng.integers(0, 3, size=len(ids)), kind="stable")  # three interleaved blocks, rank order kept inside a query
    got = idx._early_stopping_scores(ids[order], q_no[order], sp[order], qvecs, 10, 0.2, (20, 50, 100))
    want = host_statement(ids[order], q_no[order], sp[order], qvecs, 10, 0.2, (20, 50, 100))
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(bits(got[~np.isnan(got)]), bits(want[~np.isnan(want)]))
    idx.close()


def test_partial_ids_fixture_all_modes():
    z = np.load(GOLDEN / "partial_ids.npz")
    dim, seed, scale = int(z["dim"]), int(z["seed"]), float(z["scale"])
    doc_ids = [d or None for d in z["doc_ids"].tolist()]
    psg_ids = [p or None for p in z["psg_ids"].tolist()]
    vectors = synth.rows(seed, scale, 0, len(doc_ids), dim)
    qvecs = synth.rows(seed + 1000, scale, 0, 3, dim)
    queries = {"qa": "0", "qb": "1", "qc": "2"}
    for mname, mno in MODE_NO.items():
        idx = InMemoryIndex(LambdaEncoder(lambda t: qvecs[int(t)]), mode=Mode(mno), init_size=4, alloc_size=2)
        idx.add(vectors[:5], doc_ids=doc_ids[:5], psg_ids=psg_ids[:5])
        idx.add(vectors[5:9], doc_ids=doc_ids[5:9], psg_ids=psg_ids[5:9])
        idx.add(vectors[9:], doc_ids=doc_ids[9:], psg_ids=psg_ids[9:])
        run = json.loads(str(z["psg_run" if mname == "PASSAGE" else "doc_run"]))
        got = idx(Ranking.from_run(run, queries=queries))._df
        want = {(q, i): s for q, i, s in zip(z[f"{mname}_ff_q_id"].tolist(), z[f"{mname}_ff_id"].tolist(), z[f"{mname}_ff_score"])}
        have = {(q, i): s for q, i, s in zip(got["q_id"], got["id"], got["score"])}
        assert have.keys() == want.keys()
        for k in want:
            assert np.float32(have[k]).view(np.uint32) == np.float32(want[k]).view(np.uint32), (mname, k)


@pytest.mark.parametrize("name,pipelined", [("config1", False), ("d768", False), ("s2_sub", False), ("d768_f16", False), ("d768", True), ("s2_sub", True)])
def test_end_to_end_frames_match_reference(name, pipelined, monkeypatch):
    """first_stage -> index(first_stage) -> interpolate, compared frame against frame with the
    reference's output: identical scores (bits) and identical ordering (modulo exact ties).
    ``pipelined``: every batch takes the route of large ones (ffx_score: whole-query slices rotating through
    three plans, here with slices of ~300 candidates)."""
    if pipelined:
        monkeypatch.setattr(InMemoryIndex, "_PIPELINED_FROM", 1)
        monkeypatch.setenv("FFX_DEBUG_VARIANT", "score_slice_pairs=300")
    case = load_scoring_case(name)
    for mode in case.modes():
        qvecs = case.queries
        idx = InMemoryIndex(LambdaEncoder(lambda t: qvecs[int(t)]), mode=Mode(MODE_NO[mode]))
        idx.add(case.vectors, doc_ids=case.doc_names, psg_ids=case.psg_names)
        assert idx._store.dtype == case.vectors.dtype  # float16 vectors stay float16, as in the reference
        pair_off, _, sparse, names = case.pairs(mode)
        pair_q = np.repeat(np.arange(len(case.q_names)), np.diff(pair_off))
        df = pd.DataFrame({"q_id": [case.q_names[q] for q in pair_q], "id": names, "score": sparse})
        first = Ranking(df, queries={q: str(i) for i, q in enumerate(case.q_names)})
        dense = idx(first)
        exp = case.expected_order(f"{mode}_ff")
        for qn in case.q_names:
            rows = dense._df[dense._df["q_id"] == qn]
            assert_same_order_modulo_ties(rows["id"].tolist(), rows["score"].to_numpy(), exp[qn])
        assert dense._df["q_id"].tolist() == case.z[f"{mode}_ff_q_id"].tolist()
        for alpha in ALPHAS:
            exp_i = case.expected_order(f"{mode}_int{alpha}")
            host = first.interpolate(dense, alpha)
            fused = idx.rerank(first, alpha)
            for got in (host, fused):
                assert got._df["q_id"].tolist() == case.z[f"{mode}_int{alpha}_q_id"].tolist()
                for qn in case.q_names:
                    rows = got._df[got._df["q_id"] == qn]
                    assert_same_order_modulo_ties(rows["id"].tolist(), rows["score"].to_numpy(), exp_i[qn])
        top = idx.rerank(first, 0.2, cutoff=10)
        assert top == first.interpolate(dense, 0.2).cut(10)
        idx.close()


def test_fast_sum_mode_within_tolerance():
    case = load_scoring_case("d768")
    qvecs = case.queries
    idx = InMemoryIndex(LambdaEncoder(lambda t: qvecs[int(t)]), mode=Mode.MAXP, exact_sum=False)
    idx.add(case.vectors, doc_ids=case.doc_names)
    pair_off, _, sparse, names = case.pairs("MAXP")
    pair_q = np.repeat(np.arange(len(case.q_names)), np.diff(pair_off))
    df = pd.DataFrame({"q_id": [case.q_names[q] for q in pair_q], "id": names, "score": sparse})
    got = idx(Ranking(df, queries={q: str(i) for i, q in enumerate(case.q_names)}))._df
    want = case.expected("MAXP_ff")
    err = max(abs(s - want[(q, i)]) for q, i, s in zip(got["q_id"], got["id"], got["score"]))
    assert 0 < err <= 1e-4 * max(1.0, max(abs(v) for v in want.values()))  # 1e-4: BASELINE.json north_star
    assert idx.last_stats["kernel_variant"] // 100 in (2, 5, 6)  # gather_dot_fast_kernel: from a row list, or rows derived in the kernel


# ---- quantized indexes (reference tests/test_index.py:389-399 + golden frames) ------------------------
from _cases import PQ_CASES, load_pq_case  # noqa: E402
from fast_forward_amd.quantizer import NanoOPQ, NanoPQ, Quantizer  # noqa: E402


def test_quantized_index_shapes():
    """reference test_quantization: internal dim = code length, iteration yields decoded vectors."""
    pq = NanoPQ(2, 8)
    pq.fit(np.random.default_rng(0).normal(size=(16, 16)).astype(np.float32))
    idx = InMemoryIndex(quantizer=pq, mode=Mode.MAXP)
    idx.add(np.random.default_rng(1).normal(size=(5, 16)).astype(np.float32), doc_ids=DOC_IDS)
    assert idx.dim == 16 and idx._get_internal_dim() == 2
    for vec, _, _ in idx:
        assert vec.shape[0] == 16
    codes, owners = idx._get_vectors(sorted(set(DOC_IDS)))
    assert codes.shape == (5, 2) and codes.dtype == np.uint8 and owners == ["d0", "d0", "d1", "d2", "d3"]
    with pytest.raises(ValueError):
        idx.add(np.zeros((1, 8), dtype=np.float32), doc_ids=["x"])  # wrong dimensionality


@pytest.mark.parametrize("name", PQ_CASES)
def test_quantized_index_frames_match_reference(name):
    case = load_pq_case(name)
    qvecs = case.queries
    for mode in case.modes():
        idx = InMemoryIndex(LambdaEncoder(lambda t: qvecs[int(t)]), quantizer=NanoPQ.from_codewords(case.codewords), mode=Mode(MODE_NO[mode]),
                            init_size=8, alloc_size=8)
        half = len(case.raw_vectors) // 2  # two add() calls: the store grows and keeps its codebook
        idx.add(case.raw_vectors[:half], doc_ids=case.doc_names[:half], psg_ids=case.psg_names[:half])
        idx.add(case.raw_vectors[half:], doc_ids=case.doc_names[half:], psg_ids=case.psg_names[half:])
        assert idx._store.dtype == np.uint8 and np.array_equal(idx._store.gather(np.arange(len(case.codes), dtype=np.int32)), case.codes)
        pair_off, _, sparse, names = case.pairs(mode)
        pair_q = np.repeat(np.arange(len(case.q_names)), np.diff(pair_off))
        first = Ranking(pd.DataFrame({"q_id": [case.q_names[q] for q in pair_q], "id": names, "score": sparse}),
                        queries={q: str(i) for i, q in enumerate(case.q_names)})
        dense = idx(first)
        exp = case.expected_order(f"{mode}_ff")
        for qn in case.q_names:
            rows = dense._df[dense._df["q_id"] == qn]
            assert_same_order_modulo_ties(rows["id"].tolist(), rows["score"].to_numpy(), exp[qn])
        exp_i = case.expected_order(f"{mode}_int0.1")
        fused = idx.rerank(first, 0.1)
        for qn in case.q_names:
            rows = fused._df[fused._df["q_id"] == qn]
            assert_same_order_modulo_ties(rows["id"].tolist(), rows["score"].to_numpy(), exp_i[qn])
        idx.close()


def test_opq_and_generic_quantizers_score_decoded_rows():
    """OPQ: rows stay codes, queries are rotated (tolerance).  A quantizer the device cannot read:
    decoded rows are stored and scored bit-exactly."""
    import ffx_oracle as fo

    rng = np.random.default_rng(5)
    x = rng.normal(size=(400, 64)).astype(np.float32)
    qv = rng.normal(size=(3, 64)).astype(np.float32)
    doc_ids = [f"D{i}" for i in range(400)]
    run = {f"q{q}": {f"D{i}": 1.0 for i in rng.choice(400, size=50, replace=False)} for q in range(3)}
    first = Ranking.from_run(run, queries={f"q{q}": str(q) for q in range(3)})

    def oracle_scores(decoded):
        want = {}
        for q in range(3):
            for d in run[f"q{q}"]:
                want[(f"q{q}", d)] = fo.dot_f32(qv[q], decoded[int(d[1:])])
        return want

    opq = NanoOPQ(4, 16)
    opq.fit(x, pq_iter=2, rotation_iter=3)
    idx = InMemoryIndex(LambdaEncoder(lambda t: qv[int(t)]), quantizer=opq)
    idx.add(x, doc_ids=doc_ids)
    assert idx._store.dtype == np.uint8
    got = idx(first)._df
    want = oracle_scores(opq.decode(opq.encode(x)).astype(np.float32))
    err = max(abs(s - want[(q, i)]) for q, i, s in zip(got["q_id"], got["id"], got["score"]))
    assert err <= 1e-4 * max(1.0, max(abs(v) for v in want.values()))

    wide = NanoPQ(4, 300)  # uint16 codes: not a device codebook
    wide.fit(x, iter=2)
    idx2 = InMemoryIndex(LambdaEncoder(lambda t: qv[int(t)]), quantizer=wide)
    idx2.add(x, doc_ids=doc_ids)
    assert idx2._store.dtype == np.float32 and idx2._get_vectors(["D7"])[0].dtype == np.uint16
    got = idx2(first)._df
    want = oracle_scores(wide.decode(wide.encode(x)))
    for q, i, s in zip(got["q_id"], got["id"], got["score"]):
        assert np.float32(s).view(np.uint32) == np.float32(want[(q, i)]).view(np.uint32)
    idx.close()
    idx2.close()


# ---- file-to-file path and PyTerrier adapters ----------------------------------
def _write_run(path, case, pair_off, names, sparse, tag="bm25"):
    pair_q = np.repeat(np.arange(len(case.q_names)), np.diff(pair_off))
    with open(path, "w") as f:
        for q, n, s in zip(pair_q, names, sparse):
            f.write(f"{case.q_names[q]} Q0 {n} 0 {float(s)!r} {tag}\n")


@pytest.mark.parametrize("mode", ["MAXP", "PASSAGE"])
def test_rerank_file_equals_the_readme_pipeline(tmp_path, mode):
    """run file -> run file natively == Ranking.from_file -> cut -> index -> interpolate -> save."""
    case = load_scoring_case("config1")
    qvecs = case.queries
    idx = InMemoryIndex(LambdaEncoder(lambda t: qvecs[int(t)]), mode=Mode(MODE_NO[mode]))
    idx.add(case.vectors, doc_ids=case.doc_names, psg_ids=case.psg_names)
    pair_off, _, sparse, names = case.pairs(mode)
    queries = {q: str(i) for i, q in enumerate(case.q_names)}
    run_in = tmp_path / "in.tsv"
    _write_run(run_in, case, pair_off, names, sparse)
    for cutoff in (None, 30):
        first = Ranking.from_file(run_in).attach_queries(queries)
        if cutoff:
            first = first.cut(cutoff)
        want = tmp_path / f"want_{cutoff}.tsv"
        first.interpolate(idx(first), 0.1).save(want)
        got = tmp_path / "sub" / f"got_{cutoff}.tsv"
        stats = idx.rerank_file(run_in, got, 0.1, queries, cutoff=cutoff)
        assert got.read_text() == want.read_text()
        assert stats["n_queries"] == len(case.q_names) and stats["ms_gather_dot"] > 0
    with pytest.raises(ValueError, match="Queries are incomplete"):
        idx.rerank_file(run_in, tmp_path / "x.tsv", 0.1, {"q000": "0"})
    bad = tmp_path / "bad.tsv"
    bad.write_text("q000 Q0 nope 1 1.0 t\n")
    with pytest.raises(IndexError, match="ID nope not found in the index."):
        idx.rerank_file(bad, tmp_path / "x.tsv", 0.1, queries)
    idx.close()


def test_pyterrier_transformers():
    """reference util/pyterrier.py:26-48, :75-83 on plain frames (python-terrier itself is absent)."""
    from fast_forward_amd.pyterrier import FFInterpolate, FFScore

    idx = both_ids_index()
    inp = pd.DataFrame([{"qid": q, "query": text, "docno": d, "score": np.float32(s)}
                        for q, text in QUERIES.items() for d, s in DOC_RUN[q].items()])
    scored = FFScore(idx).transform(inp)
    assert set(scored.columns) >= {"qid", "docno", "score", "score_0", "query", "rank"}
    row = scored[(scored.qid == "q1") & (scored.docno == "d0")].iloc[0]
    assert row["score"] == 2.0 and row["score_0"] == 100.0
    best = scored[scored["rank"] == 0].set_index("qid")["docno"].to_dict()
    assert best == {"q1": "d3", "q2": "d3"}
    mixed = FFInterpolate(0.5).transform(scored)
    row = mixed[(mixed.qid == "q2") & (mixed.docno == "d3")].iloc[0]
    assert row["score"] == 0.5 * 800 + 0.5 * 5 and row["rank"] == 0
    assert "FFScore(" in repr(FFScore(idx))


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_index_api_matches_reference_frames(world):
    """ShardedInMemoryIndex with 2, 3 and 8 ranks (sharing this GPU; the box admits six processes to it, so eight ranks are four
    processes of two ranks, a thread each; exchange through libffx's shared-memory transport): gathered dense frames and
    per-owner interpolated frames equal the reference's."""
    from pathlib import Path

    from test_sharded_native import run_ranks

    root = Path(__file__).resolve().parent.parent
    codes, outs = run_ranks(world, root / "tests" / "_sharded_api_worker.py", ranks_per_process=2 if world > 5 else 1)
    assert codes == [0] * len(codes), "\n".join(o[0][-1500:] + o[1][-3000:] for o in outs)
    assert f"world={world}:" in outs[0][0] and "owned rows checked" in outs[0][0]


def test_call_on_a_ranking_the_user_declared_sorted_still_comes_back_sorted():
    """A Ranking built with is_sorted=True whose queries are contiguous but NOT in q_id-descending
    order: the reference re-sorts the result (base.py:463-469), so must the fast paths."""
    rng = np.random.default_rng(5)
    n, d = 60, 32
    rows = rng.standard_normal((n, d)).astype(np.float32)
    qv = rng.standard_normal((3, d)).astype(np.float32)
    index = InMemoryIndex(LambdaEncoder(lambda t: qv[int(t)]), mode=Mode.PASSAGE)
    index.add(rows, psg_ids=[f"p{i}" for i in range(n)])
    frame = pd.DataFrame({"q_id": ["qa"] * 5 + ["qc"] * 5 + ["qb"] * 5,               # ascending-ish: not the constructor's order
                          "id": [f"p{i}" for i in rng.choice(n, 15, replace=False)], "score": np.linspace(5, 1, 15).astype(np.float32),
                          "query": ["0"] * 5 + ["2"] * 5 + ["1"] * 5})
    declared = Ranking(frame, is_sorted=True)
    honest = Ranking(frame, is_sorted=False)
    for got, want in ((index(declared), index(honest)), (index.rerank(declared, 0.3), index.rerank(honest, 0.3)),
                      (index(declared, early_stopping=2, early_stopping_alpha=0.3, early_stopping_depths=(2, 5)),
                       index(honest, early_stopping=2, early_stopping_alpha=0.3, early_stopping_depths=(2, 5)))):
        assert got == want
        assert got._df["q_id"].tolist() == want._df["q_id"].tolist() == sorted(want._df["q_id"].tolist(), reverse=True)
        assert got._df["id"].tolist() == want._df["id"].tolist()
    index.close()


def test_coded_results_equal_frame_results(monkeypatch):
    """Every ranking travels through the index as integer codes and comes back coded (Ranking._assemble_coded; the frame
    of strings is built on first use), whether it holds a coded form itself (the native constructor) or a pandas frame
    (the general constructor): same rows, scores and order either way -- for __call__, rerank (with and without cutoff),
    early stopping, a repeated call (cached candidate list), and an unknown id named exactly as before."""

    rng = np.random.default_rng(12)
    n, d, n_q, k = 3000, 64, 9, 700
    rows = rng.standard_normal((n, d)).astype(np.float32)
    qv = rng.standard_normal((n_q, d)).astype(np.float32)
    index = InMemoryIndex(LambdaEncoder(lambda t: qv[int(t)]), mode=Mode.MAXP)
    index.add(rows, doc_ids=[f"D{i // 2}" for i in range(n)], psg_ids=[f"P{i}" for i in range(n)])
    frame = pd.DataFrame({"q_id": np.repeat([f"q{i}" for i in range(n_q)], k),
                          "id": np.concatenate([[f"D{x}" for x in rng.choice(n // 2, size=k, replace=False)] for _ in range(n_q)]),
                          "score": (rng.random(n_q * k) * 20).astype(np.float32)})
    first = Ranking(frame, queries={f"q{i}": str(i) for i in range(n_q)})
    assert first._coded is not None and first._frame is None
    coded = {"call": index(first), "rerank": index.rerank(first, 0.3), "cut": index.rerank(first, 0.3, cutoff=10),
             "again": index.rerank(first, 0.3), "es": index(first, early_stopping=5, early_stopping_alpha=0.3, early_stopping_depths=(5, 50, 200))}
    assert coded["rerank"]._frame is None and coded["rerank"]._coded is not None  # nothing materialised yet
    monkeypatch.setattr(Ranking, "_init_from_columns", lambda self, *a: False)  # the general constructor: a ranking that holds a frame
    fresh = Ranking(frame, queries={f"q{i}": str(i) for i in range(n_q)})
    assert fresh._coded is None and fresh._frame is not None
    plain = {"call": index(fresh), "rerank": index.rerank(fresh, 0.3), "cut": index.rerank(fresh, 0.3, cutoff=10),
             "again": index.rerank(fresh, 0.3), "es": index(fresh, early_stopping=5, early_stopping_alpha=0.3, early_stopping_depths=(5, 50, 200))}
    for key in coded:
        a, b = coded[key], plain[key]
        pd.testing.assert_frame_equal(a._df, b._df)
        assert a.q_ids == b.q_ids and a.name == b.name and a.has_queries == b.has_queries
        assert np.array_equal(a._numbering()[0], b._numbering()[0]) and a._numbering()[1] == b._numbering()[1]
    assert len(coded["cut"]._df) == n_q * 10
    # a coded ranking is a Ranking: it goes back into the index and through the host algebra
    monkeypatch.undo()
    again = index(coded["rerank"])
    assert again == index(plain["rerank"]) and first.interpolate(coded["call"], 0.3) == plain["rerank"]
    # first.interpolate(index(first), alpha) -- the reference's own idiom -- needs no join: same rows in the same order as rerank
    two_calls = first.interpolate(index(first), 0.3)
    assert two_calls._coded is not None
    pd.testing.assert_frame_equal(two_calls._df, coded["rerank"]._df)
    # ... and a dense ranking that is NOT linked to `first` (rebuilt from a frame) goes through the general join: same triples
    unlinked = Ranking(index(first)._df.copy())
    assert first.interpolate(unlinked, 0.3) == coded["rerank"]
    bad = frame.copy()
    bad.loc[4321, "id"] = "nowhere"
    bad.loc[5000, "id"] = "nowhere-either"
    broken = Ranking(bad, queries={f"q{i}": str(i) for i in range(n_q)})
    first_unknown = next(i for i in broken._df["id"] if i.startswith("nowhere"))  # in the ranking's row order
    for call in (lambda: index.rerank(broken, 0.3), lambda: index(broken)):
        with pytest.raises(IndexError, match=f"ID {first_unknown} not found in the index."):
            call()
    monkeypatch.setattr(Ranking, "_init_from_columns", lambda self, *a: False)
    with pytest.raises(IndexError, match=f"ID {first_unknown} not found in the index."):
        index.rerank(Ranking(bad, queries={f"q{i}": str(i) for i in range(n_q)}), 0.3)
    index.close()


def test_adopting_rows_that_are_already_on_the_device():
    """InMemoryIndex.adopt: ids registered over a DeviceIndex that was filled elsewhere (bench.py's
    synthetic index, a sharded load); the store stays the caller's."""
    from fast_forward_amd import native

    n, d = 5000, 128
    store = native.DeviceIndex(n, d)
    store.fill_synthetic(5, 0.1, 0, n)
    host = synth.rows(5, 0.1, 0, n, d)
    qv = synth.rows(6, 0.1, 0, 3, d)
    doc_off = np.arange(0, n + 1, 2)
    names = np.char.add("D", np.arange(n // 2).astype("U"))
    adopted = InMemoryIndex.adopt(store, doc_ids=names, doc_off=doc_off, query_encoder=LambdaEncoder(lambda t: qv[int(t)]))
    plain = InMemoryIndex(LambdaEncoder(lambda t: qv[int(t)]))
    plain.add(host, doc_ids=[f"D{i // 2}" for i in range(n)])
    rng = np.random.default_rng(2)
    run = {f"q{q}": {f"D{x}": float(rng.random()) for x in rng.choice(n // 2, size=200, replace=False)} for q in range(3)}
    first = Ranking.from_run(run, queries={f"q{q}": str(q) for q in range(3)})
    assert adopted(first) == plain(first) and len(adopted) == n and adopted.doc_ids == plain.doc_ids
    assert adopted.release() is store and len(adopted) == 0
    adopted.close()
    assert store.gather(np.array([7], dtype=np.int32)).shape == (1, d)  # still alive
    store.close()
    plain.close()


def test_a_result_keeps_its_arrays_when_later_calls_reuse_the_result_pool():
    """Index.__call__ / rerank on a large run return arrays that live in page-locked blocks of the result pool (the device
    writes the run there in its final order).  A block goes back to the pool only when the LAST view of it dies: a later
    large call on the same index must not disturb an earlier result, and a dropped result's block is what the next call
    gets.  1.2 M pairs per call (the device applies the ordering from 256 K)."""
    from fast_forward_amd import native

    rng = np.random.default_rng(5)
    n, d, n_q, k = 60_000, 32, 300, 4000
    rows = rng.standard_normal((n, d)).astype(np.float32)
    qv = rng.standard_normal((n_q, d)).astype(np.float32)
    index = InMemoryIndex(LambdaEncoder(lambda t: qv[int(t)]), mode=Mode.PASSAGE)
    index.add(rows, psg_ids=[f"P{i}" for i in range(n)])

    def run(seed):
        r = np.random.default_rng(seed)
        ids = np.concatenate([r.choice(n, size=k, replace=False) for _ in range(n_q)])
        frame = pd.DataFrame({"q_id": np.repeat([f"q{i:03d}" for i in range(n_q)], k), "id": [f"P{x}" for x in ids],
                              "score": (r.random(n_q * k) * 20).astype(np.float32)})
        return Ranking(frame, queries={f"q{i:03d}": str(i) for i in range(n_q)})

    first, second = run(1), run(2)
    dense_first = index(first)
    link = dense_first._source_rows[1]
    keep_link, keep_scores, keep_codes = link.copy(), dense_first._coded["score"].copy(), dense_first._coded["id_codes"].copy()
    dense_second = index(second)       # more blocks of the pool, not these
    for _ in range(3):
        index.rerank(second, 0.3)      # ... and results that die at once: their blocks go round
    assert np.array_equal(dense_first._source_rows[1], keep_link)
    assert np.array_equal(dense_first._coded["score"].view(np.uint32), keep_scores.view(np.uint32)) and np.array_equal(dense_first._coded["id_codes"], keep_codes)
    assert first.interpolate(dense_first, 0.2) == index.rerank(first, 0.2)
    assert second.interpolate(dense_second, 0.2) == index.rerank(second, 0.2)
    # a dropped result hands its block back: the next result of that size is served from the pool
    idle_before = native._POOL_IDLE[0]
    gone = index.rerank(second, 0.5)
    addr = gone._coded["score"].ctypes.data
    del gone
    assert native._POOL_IDLE[0] > idle_before - 1  # (nothing was lost on the way)
    again = index.rerank(second, 0.5)
    assert again._coded["score"].ctypes.data == addr or again._coded["id_codes"].ctypes.data == addr
    index.close()
    del again, dense_first, dense_second
    import gc

    gc.collect()
    assert native.drain_result_pool() > 0 and native._POOL_IDLE[0] == 0


def test_interpolate_after_call_starts_from_the_dense_scores_the_device_kept():
    """``first.interpolate(index(first), alpha)`` -- the reference's documented pipeline (README; ranking.py:293-326 after
    index/base.py:389-469) -- does not score twice: ``index(first)`` leaves its dense scores (and the ranking's id codes)
    in HBM (FFX_FLAG_KEEP_DENSE) and the interpolate is the interpolation + ordering on the device alone (ffx_rank_kept).
    Same Ranking as ``index.rerank`` for every alpha; MaxP over documents of 1 - 3 passages (the kept scores are the
    aggregated ones); once another call has displaced the kept scores the interpolate combines the dense ranking's own scores on
    the host -- same result, and nothing is encoded or scored again."""
    from fast_forward_amd import native

    rng = np.random.default_rng(11)
    n_docs, d, n_q, k = 50_000, 48, 280, 4000
    per_doc = rng.integers(1, 4, size=n_docs)
    doc_of_row = np.repeat(np.arange(n_docs), per_doc)
    rows = rng.standard_normal((len(doc_of_row), d)).astype(np.float32)
    qv = rng.standard_normal((n_q, d)).astype(np.float32)
    index = InMemoryIndex(LambdaEncoder(lambda t: qv[int(t)]), mode=Mode.MAXP)
    index.add(rows, doc_ids=[f"D{x}" for x in doc_of_row])

    def run(seed):
        r = np.random.default_rng(seed)
        ids = np.concatenate([r.choice(n_docs, size=k, replace=False) for _ in range(n_q)])
        frame = pd.DataFrame({"q_id": np.repeat([f"q{i:03d}" for i in range(n_q)], k), "id": [f"D{x}" for x in ids],
                              "score": (r.random(n_q * k) * 20).astype(np.float32)})
        return Ranking(frame, queries={f"q{i:03d}": str(i) for i in range(n_q)})

    first, second = run(1), run(2)
    dense = index(first)
    token = index._kept_token()
    assert token != 0 and dense._origin[3] == token
    for alpha in (0.0, 0.2, 0.75, 1.0):
        index.last_timings = None
        got = first.interpolate(dense, alpha)
        assert "queries_numbered_s" in index.last_timings  # the kept route (no query encoded, no row read)
        assert index._kept_token() == token                 # ... which leaves the kept scores where they are
        want = index.rerank(first, alpha)
        assert got == want and np.array_equal(bits(got._coded["score"]), bits(want._coded["score"]))
        assert np.array_equal(got._coded["id_codes"], want._coded["id_codes"])
    # C ABI: the token is bound to the candidate lists, and goes with the next keeping call
    off = np.arange(n_q + 1, dtype=np.int64) * k
    with pytest.raises(native.FFXError):
        native.rank_kept(index._store, token, off[:-1], None, 0.0)
    dense_order, codes, order, _ = native.rank_kept(index._store, token, off, None, 0.0, want_order=True)  # no first-stage scores: dense, ordered
    assert np.array_equal(bits(dense_order), bits(dense._coded["score"])) and np.array_equal(codes, dense._coded["id_codes"])
    assert np.array_equal(order, dense._source_rows[1])
    dense_second = index(second)
    assert index._kept_token() not in (0, token)
    with pytest.raises(native.FFXError):
        native.rank_kept(index._store, token, off, None, 0.0)
    # the kept scores are gone: interpolate is arithmetic on the two rankings (ranking.py:293-326) -- `dense`'s OWN scores, combined
    # and ordered on the host; no query is encoded and no candidate scored again, whatever has happened to the index since
    assert second.interpolate(dense_second, 0.2) == index.rerank(second, 0.2)
    want = index.rerank(first, 0.2)
    encoded = []
    index.query_encoder = LambdaEncoder(lambda t: encoded.append(t) or qv[int(t)] * 0)  # (an encoder that would give other scores)
    index.last_timings = None
    got = first.interpolate(dense, 0.2)
    assert not encoded and index.last_timings is None
    assert got == want and np.array_equal(bits(got._coded["score"]), bits(want._coded["score"])) and np.array_equal(got._coded["id_codes"], want._coded["id_codes"])
    index.close()


def test_early_stopping_over_documents_of_several_passages_at_768():
    """Early stopping where every interval's scoring derives its rows inside gather_dot (d = 768, documents of 1 - 4 passages, MaxP and
    AveP; the interval's candidate lists change every time, so do the tile tables): the device loop equals the NumPy statement of the
    reference's loop (``Index._early_stopping_kept``, base.py:316-387, which scores every interval through ``_score_pairs``)."""
    from fast_forward_amd.index.api import Index as BaseIndex

    rng = np.random.default_rng(21)
    n_docs, d, n_q, k = 3000, 768, 12, 700
    per_doc = rng.integers(1, 5, size=n_docs)
    doc_of_row = np.repeat(np.arange(n_docs), per_doc)
    qv = (rng.standard_normal((n_q, d)) * 0.05).astype(np.float32)
    frame = pd.DataFrame({"q_id": np.repeat([f"q{i:02d}" for i in range(n_q)], k),
                          "id": np.concatenate([[f"D{x}" for x in rng.choice(n_docs, size=k, replace=False)] for _ in range(n_q)]),
                          "score": -np.sort(-(rng.random((n_q, k)) * 3).astype(np.float32), axis=1).reshape(-1)})
    first = Ranking(frame, queries={f"q{i:02d}": str(i) for i in range(n_q)})
    for mode in (Mode.MAXP, Mode.AVEP):
        idx = InMemoryIndex(LambdaEncoder(lambda t: qv[int(t)]), mode=mode)
        idx.add((rng.standard_normal((len(doc_of_row), d)) * 0.05).astype(np.float32), doc_ids=[f"D{x}" for x in doc_of_row])
        for cutoff, alpha, depths in ((10, 0.3, (10, 100, 700)), (3, 0.1, (5, 50, 200, 700))):
            got = idx(first, early_stopping=cutoff, early_stopping_alpha=alpha, early_stopping_depths=depths)
            device_loop = InMemoryIndex._early_stopping_kept
            try:
                InMemoryIndex._early_stopping_kept = BaseIndex._early_stopping_kept  # the NumPy statement, intervals scored by the device
                want = idx(first, early_stopping=cutoff, early_stopping_alpha=alpha, early_stopping_depths=depths)
            finally:
                InMemoryIndex._early_stopping_kept = device_loop
            assert got == want and 0 < len(got._df) < len(frame), (mode, cutoff)
            assert idx.last_stats["kernel_variant"] in (311, 411), idx.last_stats  # (411: an interval whose candidates all have one row)
        idx.close()


def test_early_stopping_on_a_large_run_edge_cases():
    """The coded early-stopping route (>= 4096 rows: id codes in, only the scored candidates back, a coded Ranking out)
    on the edges: depths all below the cutoff (nothing is scored: an empty ranking), an unknown id that IS reached
    (IndexError naming it) and one that is not (fine, as in the reference), a cutoff beyond every list (everything
    scored once: equals the plain call)."""
    rng = np.random.default_rng(0)
    n, d, n_q, k = 5000, 32, 10, 600
    qv = rng.standard_normal((n_q, d)).astype(np.float32)
    idx = InMemoryIndex(LambdaEncoder(lambda t: qv[int(t)]), mode=Mode.PASSAGE)
    idx.add(rng.standard_normal((n, d)).astype(np.float32), psg_ids=[f"P{i}" for i in range(n)])
    frame = pd.DataFrame({"q_id": np.repeat([f"q{i}" for i in range(n_q)], k),
                          "id": np.concatenate([[f"P{x}" for x in rng.choice(n, size=k, replace=False)] for _ in range(n_q)]),
                          "score": -np.sort(-(rng.random((n_q, k)) * 20).astype(np.float32), axis=1).reshape(-1)})
    queries = {f"q{i}": str(i) for i in range(n_q)}
    first = Ranking(frame, queries=queries)
    nothing = idx(first, early_stopping=50, early_stopping_alpha=0.2, early_stopping_depths=(5, 10))
    assert nothing._coded is not None and len(nothing) == 0 and len(nothing._df) == 0
    bad = frame.copy()
    bad.loc[3, "id"] = "nowhere"
    with pytest.raises(IndexError, match="nowhere"):
        idx(Ranking(bad, queries=queries), early_stopping=5, early_stopping_alpha=0.2, early_stopping_depths=(5, 50))
    bad = frame.copy()
    bad.loc[590, "id"] = "nowhere"
    fine = idx(Ranking(bad, queries=queries), early_stopping=5, early_stopping_alpha=0.2, early_stopping_depths=(5, 50))
    assert 0 < len(fine._df) < len(frame)
    everything = idx(first, early_stopping=10**6, early_stopping_alpha=0.2, early_stopping_depths=(10**6,))
    assert everything == idx(first) and len(everything._df) == len(frame)
    idx.close()
