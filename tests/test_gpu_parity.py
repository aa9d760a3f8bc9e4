"""GPU parity: the HIP path through the C ABI (libffx.so) against the CPU oracle and the
golden fixtures produced by the reference.  Run with ``-m gpu`` on an MI355X.

Bars (BASELINE.json north_star):
  * exact-sum mode: scores BIT-IDENTICAL to the reference (hence identical run ordering)
  * fast mode: |score - reference| <= 1e-4 (fp32, unit-scale rows), tolerance written below
  * aggregation, interpolation, ordering: bit-exact / identical modulo exact ties
"""

import os

import numpy as np
import pytest

import ffx_oracle as fo
from _cases import (ALPHAS, F64_ALPHAS, F64_CASES, MODE_NO, SCORING_CASES, assert_same_order_modulo_ties, assert_within_reference_tolerance,
                    bits, load_scoring_case, load_sub200k)
from fast_forward_amd import native, synth

pytestmark = pytest.mark.gpu

FAST_TOL = 1e-4  # north_star: "match the reference NumPy scores to 1e-4 fp32"


def device_index(case) -> native.DeviceIndex:
    idx = native.DeviceIndex(case.vectors.shape[0], case.vectors.shape[1], dtype=case.vectors.dtype)
    idx.upload(case.vectors, 0)
    idx.set_doc_rows(case.doc_off, case.doc_rows)
    idx.set_psg_rows(case.psg_row)
    return idx


@pytest.fixture(scope="module")
def cases():
    return {name: load_scoring_case(name) for name in SCORING_CASES}


@pytest.mark.parametrize("name", SCORING_CASES)
def test_exact_scores_bit_identical_to_reference(cases, name):
    case = cases[name]
    idx = device_index(case)
    for mode in case.modes():
        pair_off, pair_ids, sparse, names = case.pairs(mode)
        ff, interp, order, st = native.score(idx, case.queries, pair_off, pair_ids, MODE_NO[mode], sparse, 0.1,
                                             exact_sum=True, sort=True)
        exp = case.expected(f"{mode}_ff")
        pair_q = np.repeat(np.arange(len(case.q_names)), np.diff(pair_off))
        want = np.array([exp[(case.q_names[q], n)] for q, n in zip(pair_q, names)], dtype=np.float32)
        assert np.array_equal(bits(ff), bits(want)), f"{name}/{mode}: max diff {np.abs(ff - want).max()} (variant {st['kernel_variant']})"
        exp_i = case.expected(f"{mode}_int0.1")
        want_i = np.array([exp_i[(case.q_names[q], n)] for q, n in zip(pair_q, names)], dtype=np.float32)
        assert np.array_equal(bits(interp), bits(want_i))
        # device ordering == reference ordering (modulo exact ties)
        exp_order = case.expected_order(f"{mode}_int0.1")
        for q, qn in enumerate(case.q_names):
            sel = order[pair_off[q]:pair_off[q + 1]]
            assert_same_order_modulo_ties([names[i] for i in sel], interp[sel], exp_order[qn])
        # and == the oracle's deterministic tie-break (stable: ascending position)
        assert np.array_equal(order.astype(np.int64), fo.order(interp, pair_off))
    idx.close()


@pytest.mark.parametrize("name", SCORING_CASES)
def test_fast_scores_within_tolerance(cases, name):
    case = cases[name]
    idx = device_index(case)
    for mode in case.modes():
        pair_off, pair_ids, sparse, _ = case.pairs(mode)
        ff, _, _, _ = native.score(idx, case.queries, pair_off, pair_ids, MODE_NO[mode], exact_sum=False)
        want, _ = fo.score(case.vectors, case.doc_off, case.doc_rows, case.psg_row, case.queries, pair_off, pair_ids, MODE_NO[mode])
        scale = max(1.0, float(np.abs(want).max()))
        assert np.abs(ff - want).max() <= FAST_TOL * scale, f"{name}/{mode}"
    idx.close()


@pytest.mark.parametrize("exact", [True, False], ids=["exact_sum", "fast_sum"])
@pytest.mark.parametrize("name", F64_CASES)
def test_scores_within_tolerance_of_the_references_default_float64_path(name, exact, record_property):
    """The reference's DEFAULT index scores in float64 once it holds more than one chunk
    (index/memory.py:106; fixtures: the reference run with init_size = alloc_size = 64).  Both
    summation modes of the HIP path must sit within 1e-4 * max(1, |s|_inf) of those frames, and order
    the candidates identically except inside reference score gaps below 2e-4 * that scale."""
    case = load_scoring_case(name)
    idx = device_index(case)
    for mode in case.modes():
        pair_off, pair_ids, sparse, names = case.pairs(mode)
        pair_q = np.repeat(np.arange(len(case.q_names)), np.diff(pair_off))
        for alpha in F64_ALPHAS:
            ff, interp, order, _ = native.score(idx, case.queries, pair_off, pair_ids, MODE_NO[mode], sparse, alpha, exact_sum=exact, sort=True)
            exp, exp_i = case.expected(f"{mode}_ff"), case.expected(f"{mode}_int{alpha}")
            want = np.array([exp[(case.q_names[q], n)] for q, n in zip(pair_q, names)], dtype=np.float32)
            want_i = np.array([exp_i[(case.q_names[q], n)] for q, n in zip(pair_q, names)], dtype=np.float32)
            rep = assert_within_reference_tolerance(ff, want, pair_off, f"{name}/{mode}/dense")
            rep_i = assert_within_reference_tolerance(interp, want_i, pair_off, f"{name}/{mode}/alpha={alpha}")
            record_property(f"{mode}_a{alpha}", {"dense": rep, "interpolated": rep_i})
            # the device's own order is the descending order of its own scores
            for q in range(len(case.q_names)):
                sel = order[pair_off[q]:pair_off[q + 1]]
                assert (np.diff(interp[sel]) <= 0).all()
    idx.close()


def test_sub_sampled_config2_200k_rows_against_both_reference_runs(record_property):
    """BASELINE.md section 4's sub-sample (200 k rows x 768, Q = 200, k = 1000, MaxP; rows generated on the
    device from the fixture's seed): the HIP path equals the reference's single-fp32-chunk run bit
    for bit, and both summation modes sit within 1e-4 of its DEFAULT float64-chunk run with the same
    ordering outside tiny gaps."""
    c = load_sub200k()
    idx = native.DeviceIndex(c["n_rows"], c["dim"])
    idx.fill_synthetic(c["seed"], c["scale"], 0, c["n_rows"])
    idx.set_doc_rows(c["doc_off"], np.arange(c["n_rows"], dtype=np.int32))
    idx.set_psg_rows(np.arange(c["n_rows"], dtype=np.int32))
    ff, interp, order, st = native.score(idx, c["queries"], c["pair_off"], c["pair_ids"], MODE_NO["MAXP"], c["sparse"], 0.1, exact_sum=True, sort=True)
    assert st["n_rows"] > 4 * st["n_pairs"]  # 1..8 passages per document
    assert np.array_equal(bits(ff), bits(c["ff_f32chunk"]))
    assert np.array_equal(bits(interp), bits(fo.interpolate(c["sparse"], c["ff_f32chunk"], 0.1)))
    assert np.array_equal(order.astype(np.int64), fo.order(interp, c["pair_off"]))
    record_property("exact_vs_float64_chunks", assert_within_reference_tolerance(ff, c["ff_default_chunks"], c["pair_off"], "exact"))
    fast, _, _, _ = native.score(idx, c["queries"], c["pair_off"], c["pair_ids"], MODE_NO["MAXP"], exact_sum=False)
    record_property("fast_vs_float64_chunks", assert_within_reference_tolerance(fast, c["ff_default_chunks"], c["pair_off"], "fast"))
    record_property("fast_vs_fp32_chunk", assert_within_reference_tolerance(fast, c["ff_f32chunk"], c["pair_off"], "fast vs fp32"))
    idx.close()


@pytest.mark.parametrize("alpha", ALPHAS)
def test_interpolation_all_alphas(cases, alpha):
    case = cases["d768"]
    idx = device_index(case)
    plan = native.Plan(idx)
    plan.set_queries(case.queries)
    for mode in case.modes():
        pair_off, pair_ids, sparse, names = case.pairs(mode)
        plan.set_pairs(pair_off, pair_ids, sparse, MODE_NO[mode])
        plan.run(alpha, exact_sum=True)
        _, interp, _ = plan.fetch(interp=True)
        exp = case.expected(f"{mode}_int{alpha}")
        pair_q = np.repeat(np.arange(len(case.q_names)), np.diff(pair_off))
        want = np.array([exp[(case.q_names[q], n)] for q, n in zip(pair_q, names)], dtype=np.float32)
        assert np.array_equal(bits(interp), bits(want))
    plan.close()
    idx.close()


def test_plan_rerun_and_stats(cases):
    case = cases["s2_sub"]
    idx = device_index(case)
    plan = native.Plan(idx)
    plan.set_queries(case.queries)
    pair_off, pair_ids, sparse, _ = case.pairs("MAXP")
    plan.set_pairs(pair_off, pair_ids, sparse, MODE_NO["MAXP"])
    outs = []
    for alpha in (0.1, 0.5, 0.1):
        plan.run(alpha, exact_sum=True, sort=True)
        outs.append(plan.fetch(interp=True, order=True))
    assert np.array_equal(outs[0][0], outs[2][0]) and np.array_equal(outs[0][1], outs[2][1])
    assert np.array_equal(outs[0][2], outs[2][2])
    assert not np.array_equal(outs[0][1], outs[1][1])
    st = plan.stats()
    d = case.vectors.shape[1]
    assert st["n_pairs"] == pair_ids.shape[0] and st["n_rows"] > st["n_pairs"]
    want_bytes = st["n_rows"] * (d * 4 + 4) + st["n_pairs"] * 12 + st["n_queries"] * d * 4 + (st["n_queries"] + 1) * 8
    assert st["algorithmic_bytes"] == want_bytes
    assert st["ms_gather_dot"] > 0
    plan.close()
    idx.close()


def test_unknown_ordinal_is_not_found(cases):
    case = cases["dim20"]
    idx = device_index(case)
    pair_off, pair_ids, _, _ = case.pairs("MAXP")
    bad = pair_ids.copy()
    bad[5] = 10**6
    bad[9] = -3
    with pytest.raises(native.NotFound) as e:
        native.score(idx, case.queries, pair_off, bad, MODE_NO["MAXP"])
    assert e.value.pair == 5
    idx.close()


def test_early_stopping_native_edge_cases(cases):
    """ffx_plan_early_stopping on degenerate inputs: no queries, no candidates, depths all below the cutoff
    (nothing is scored), a cutoff beyond every list, an unresolved ordinal (-1) that is never reached (fine)
    and one that is reached in the second interval (NotFound naming its position)."""
    case = cases["dim20"]
    idx = device_index(case)
    plan = native.Plan(idx)
    d = case.queries.shape[1]
    # no queries / no candidates
    plan.set_queries(np.zeros((0, d), dtype=np.float32))
    plan.set_pairs_lazy(np.array([0]), np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.float32), 2)
    assert plan.early_stopping(3, 0.5, (5, 10)) == (0, 0)
    plan.set_queries(case.queries[:2])
    plan.set_pairs_lazy(np.array([0, 0, 0]), np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.float32), 2)
    assert plan.early_stopping(3, 0.5, (5, 10)) == (0, 0)
    # two queries, 9 and 4 candidates
    n_docs = len(case.doc_off) - 1
    ids = (np.arange(13) % n_docs).astype(np.int32)
    sparse = np.linspace(9, 1, 13).astype(np.float32)
    off = np.array([0, 9, 13])
    plan.set_pairs_lazy(off, ids, sparse, 2)
    assert plan.early_stopping(5, 0.5, (2, 3, 4)) == (0, 0)              # every depth below the cutoff: nothing scored
    assert np.isnan(plan.fetch()[0]).all()
    scored, intervals = plan.early_stopping(100, 0.5, (100, 200))          # cutoff beyond every list: one interval takes all
    assert (scored, intervals) == (13, 1) and not np.isnan(plan.fetch()[0]).any()
    want, _ = fo.score(case.vectors, case.doc_off, case.doc_rows, case.psg_row, case.queries[:2], off, ids, 2)
    assert np.array_equal(bits(plan.fetch()[0]), bits(want))
    # an unresolved id below the stopping depth is never looked at
    lazy = ids.copy()
    lazy[8] = -1
    plan.set_pairs_lazy(off, lazy, np.concatenate([np.linspace(90, 10, 9), np.linspace(9, 1, 4)]).astype(np.float32), 2)
    scored, intervals = plan.early_stopping(2, 1.0, (2, 20))  # alpha = 1: the bound is the last sparse score, far below the top-2
    assert intervals == 1 and scored == 4                      # ranks 0..1 of both queries; nobody continues
    # ... but reached in the second interval it is the reference's IndexError
    plan.set_pairs_lazy(off, lazy, np.full(13, 1.0, dtype=np.float32), 2)
    with pytest.raises(native.NotFound) as e:
        plan.early_stopping(2, 0.0, (2, 20))                   # alpha = 0: everybody continues (best dense > k-th best unless equal)
    assert e.value.pair == 8
    plan.close()
    idx.close()


def test_query_vectors_handed_over_in_device_memory(cases):
    """ffx_plan_set_queries_device: query vectors that already sit in HBM (a GPU-side encoder's output; here:
    the rows of a second index on the same device, via ffx_index_device_ptr) give the same bits as the
    same vectors uploaded from the host."""
    case = cases["d768"]
    idx = device_index(case)
    holder = native.DeviceIndex(case.queries.shape[0], case.queries.shape[1])
    holder.upload(case.queries, 0)
    pair_off, pair_ids, sparse, _ = case.pairs("MAXP")
    a, b = native.Plan(idx), native.Plan(idx)
    a.set_queries(case.queries)
    b.set_queries_device(holder.device_ptr(), case.queries.shape[0])
    outs = []
    for plan in (a, b):
        plan.set_pairs(pair_off, pair_ids, sparse, MODE_NO["MAXP"])
        plan.run(0.1, exact_sum=True, sort=True)
        outs.append(plan.fetch(interp=True, order=True))
    for x, y in zip(*outs):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    with pytest.raises(native.FFXError):
        b.set_queries_device(0, 3)
    idx.close()
    holder.close()


def test_closing_an_index_closes_what_was_built_on_it(cases):
    """A Plan (or sharded scorer) holds device state that refers to its index: closing the index first
    closes them, so their own close / collection later is a no-op instead of a use-after-free."""
    case = cases["dim20"]
    idx = device_index(case)
    plan, other = native.Plan(idx), native.Plan(idx)
    pair_off, pair_ids, sparse, _ = case.pairs("MAXP")
    plan.set_queries(case.queries)
    plan.set_pairs(pair_off, pair_ids, sparse, MODE_NO["MAXP"])
    plan.run(0.1)
    idx.close()
    assert plan._h is None and other._h is None
    plan.close()
    del plan, other


def test_row_count_beyond_int32_is_refused_not_wrapped():
    """The row list is addressed with int32.  A batch that would gather more than 2^31-1 rows is refused --
    also when the int32 scan of the per-candidate counts would wrap past 2^32 back into positive
    numbers (62 000 candidates x a 70 000-passage document = 4.34e9 rows)."""
    n, d = 70_000, 8
    idx = native.DeviceIndex(n, d)
    idx.fill_synthetic(1, 1.0, 0, n)
    idx.set_doc_rows(np.array([0, n]), np.arange(n))  # ONE document owning every row
    idx.set_psg_rows(np.arange(n))
    plan = native.Plan(idx)
    plan.set_queries(np.ones((2, d), dtype=np.float32))
    for p in (31_000, 62_000):  # 2.17e9 (wraps negative) and 4.34e9 (wraps to a positive number)
        with pytest.raises(native.FFXError, match="gathered rows in one batch"):
            plan.set_pairs(np.array([0, p // 2, p]), np.zeros(p, dtype=np.int32), None, MODE_NO["MAXP"])
    plan.set_pairs(np.array([0, 2, 3]), np.zeros(3, dtype=np.int32), None, MODE_NO["MAXP"])  # 210 000 rows: fine
    plan.run(0.0)
    assert plan.stats()["n_rows"] == 3 * n
    plan.set_pairs(np.array([0, 31_000, 62_000]), np.zeros(62_000, dtype=np.int32), None, MODE_NO["FIRSTP"])  # one row each: fine
    plan.close()
    idx.close()


def test_ragged_and_empty_queries():
    """Queries with 0, 1 and many candidates; a tile boundary inside a query; R not a multiple
    of the wave tile."""
    d, n = 768, 700
    vectors = synth.rows(11, 0.05, 0, n, d)
    queries = synth.rows(12, 0.05, 0, 6, d)
    idx = native.DeviceIndex(n, d)
    idx.upload(vectors, 0)
    doc_off = np.arange(n + 1)
    idx.set_doc_rows(doc_off, np.arange(n))
    idx.set_psg_rows(np.arange(n))
    lens = [0, 1, 333, 0, 17, 64]
    pair_off = np.concatenate([[0], np.cumsum(lens)])
    rng = np.random.default_rng(0)
    pair_ids = np.concatenate([rng.choice(n, size=k, replace=False) for k in lens]).astype(np.int32)
    for mode in (1, 2, 3, 4):
        ff, _, order, _ = native.score(idx, queries, pair_off, pair_ids, mode, sort=True)
        want, _ = fo.score(vectors, doc_off, np.arange(n), np.arange(n), queries, pair_off, pair_ids, mode)
        assert np.array_equal(bits(ff), bits(want))
        assert np.array_equal(order.astype(np.int64), fo.order(want, pair_off))
    # nothing to score at all
    ff, _, _, _ = native.score(idx, queries, np.zeros(7, dtype=np.int64), np.zeros(0, dtype=np.int32), 2)
    assert ff.shape == (0,)
    idx.close()


def test_single_submission_path_matches_general_path(monkeypatch):
    """ffx_score on a small batch of single-row candidates queues uploads, validation, kernels and downloads
    in one go and waits once; its results are those of the general path (FFX_DEBUG_VARIANT=no_small_path) bit for bit, for
    every mode, with and without first-stage scores and ordering, for non-identity id -> row tables and for
    fp16 rows; a bad ordinal (too large, negative, or a passage id whose row is gone) is the same NotFound."""
    d, n = 768, 900
    vectors = synth.rows(21, 0.05, 0, n, d)
    queries = synth.rows(22, 0.05, 0, 5, d)
    rng = np.random.default_rng(3)
    perm = rng.permutation(n).astype(np.int32)  # ordinal o lives in row perm[o]
    lens = [0, 1, 700, 130, 64]
    pair_off = np.concatenate([[0], np.cumsum(lens)])
    pair_ids = np.concatenate([rng.choice(n, size=k, replace=False) for k in lens]).astype(np.int32)
    sparse = rng.normal(size=pair_ids.shape[0]).astype(np.float32)
    for dtype in ("f32", "f16"):
        idx = native.DeviceIndex(n, d, dtype=np.float16 if dtype == "f16" else np.float32)
        idx.upload(vectors.astype(np.float16) if dtype == "f16" else vectors, 0)
        idx.set_doc_rows(np.arange(n + 1), perm)
        idx.set_psg_rows(perm)
        for mode in (1, 2, 3, 4):
            for sp, sort in ((None, False), (sparse, True), (None, True)):
                monkeypatch.delenv("FFX_DEBUG_VARIANT", raising=False)
                a = native.score(idx, queries, pair_off, pair_ids, mode, sp, 0.3, sort=sort)
                monkeypatch.setenv("FFX_DEBUG_VARIANT", "no_small_path")
                b = native.score(idx, queries, pair_off, pair_ids, mode, sp, 0.3, sort=sort)
                for x, y in zip(a[:3], b[:3]):
                    assert (x is None) == (y is None)
                    if x is not None:
                        assert np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y)
                assert a[3]["n_pairs"] == b[3]["n_pairs"] and a[3]["n_rows"] == b[3]["n_rows"] and a[3]["kernel_variant"] == b[3]["kernel_variant"]
        monkeypatch.delenv("FFX_DEBUG_VARIANT", raising=False)
        for value, at in ((n, 7), (-1, 0), (2**31 - 1, pair_ids.shape[0] - 1)):
            bad = pair_ids.copy()
            bad[at] = value
            bad[-1] = value if at != pair_ids.shape[0] - 1 else bad[-1]
            for mode in (1, 2, 3, 4):
                with pytest.raises(native.NotFound) as e:
                    native.score(idx, queries, pair_off, bad, mode, sparse, 0.3, sort=True)
                assert e.value.pair == at
        # the call after a refused one is unaffected
        ok = native.score(idx, queries, pair_off, pair_ids, 4, None, 0.3, sort=True)
        assert np.array_equal(ok[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(ok[2], b[2])
        idx.close()
    # a passage id whose row was dropped (psg_row = -1)
    idx = native.DeviceIndex(n, d)
    idx.upload(vectors, 0)
    holes = perm.copy()
    holes[0] = -1
    holes[11] = -1
    idx.set_doc_rows(np.arange(n + 1), perm)
    idx.set_psg_rows(holes)
    bad = pair_ids.copy()
    bad[3] = 11
    with pytest.raises(native.NotFound) as e:
        native.score(idx, queries, pair_off, bad, 1)
    assert e.value.pair == 3
    idx.close()


def test_many_passages_per_document():
    """MAXP / AVEP segments from 1 to 300 rows, rows of a document NOT contiguous."""
    d, n_docs = 128, 40
    rng = np.random.default_rng(3)
    per_doc = rng.integers(1, 12, size=n_docs)
    per_doc[7] = 300
    n = int(per_doc.sum())
    doc_of_row = rng.permutation(np.repeat(np.arange(n_docs), per_doc))
    doc_rows = np.argsort(doc_of_row, kind="stable").astype(np.int32)
    doc_off = np.concatenate([[0], np.cumsum(np.bincount(doc_of_row, minlength=n_docs))])
    vectors = synth.rows(21, 0.3, 0, n, d)
    queries = synth.rows(22, 0.3, 0, 5, d)
    idx = native.DeviceIndex(n, d)
    idx.upload(vectors, 0)
    idx.set_doc_rows(doc_off, doc_rows)
    idx.set_psg_rows(np.arange(n))
    pair_off = np.arange(6) * n_docs
    pair_ids = np.concatenate([rng.permutation(n_docs) for _ in range(5)]).astype(np.int32)
    sparse = rng.random(pair_ids.shape[0]).astype(np.float32)
    for mode in (2, 3, 4):
        ff, interp, _, _ = native.score(idx, queries, pair_off, pair_ids, mode, sparse, 0.2)
        want, want_i = fo.score(vectors, doc_off, doc_rows, np.arange(n), queries, pair_off, pair_ids, mode, sparse, 0.2)
        assert np.array_equal(bits(ff), bits(want)), mode
        assert np.array_equal(bits(interp), bits(want_i)), mode
    idx.close()


@pytest.mark.parametrize("store", [np.float32, np.float16])
@pytest.mark.parametrize("d", [768, 1024])
@pytest.mark.parametrize("contiguous", [True, False])
def test_rows_derived_inside_gather_dot_for_documents_of_several_passages(monkeypatch, d, store, contiguous):
    """At d = 768 / 1024 no row list is written.  MAXP / AVEP over documents of 1 - 9 (one of 70) passages: the waves of
    gather_dot work the rows of their tile out of the candidate lists, the row offsets and two small tables (kernel variants
    311 / 312, +1000 for half rows; get_indices, index/util.py:29-41 + the merge of index/base.py:296-298); FIRSTP / PASSAGE: a
    candidate's one row is looked up inside the kernel (411 / 412).  Same bits as the
    oracle and as the row-list form (FFX_DEBUG_VARIANT=no_derive -> variants 111 / 112) -- documents whose rows are scattered
    over the index or contiguous, queries without candidates at the start, in the middle and at the end, lists of one candidate,
    a last tile that is not full; a plan that is run again, and with other candidate lists."""
    rng = np.random.default_rng(d + (1 if contiguous else 0))
    n_docs = 600
    per_doc = rng.integers(1, 10, size=n_docs)
    per_doc[11] = 70
    n = int(per_doc.sum())
    doc_of_row = np.repeat(np.arange(n_docs), per_doc)
    if not contiguous:
        doc_of_row = rng.permutation(doc_of_row)
    doc_rows = np.argsort(doc_of_row, kind="stable").astype(np.int32)
    doc_off = np.concatenate([[0], np.cumsum(np.bincount(doc_of_row, minlength=n_docs))])
    vectors = synth.rows(5, 0.05, 0, n, d).astype(store)
    as_f32 = vectors.astype(np.float32)
    lens = [0, 200, 1, 0, 0, 333, 57, 600, 0]
    queries = synth.rows(6, 0.05, 0, len(lens), d)
    idx = native.DeviceIndex(n, d, dtype=store)
    idx.upload(vectors, 0)
    idx.set_doc_rows(doc_off, doc_rows)
    idx.set_psg_rows(np.arange(n))
    plan = native.Plan(idx)
    plan.set_queries(queries)
    half = 1000 if store == np.float16 else 0
    for trial in range(2):
        pair_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        pair_ids = np.concatenate([rng.choice(n_docs, size=k, replace=False) for k in lens] + [np.zeros(0, dtype=np.int64)]).astype(np.int32)
        sparse = rng.random(len(pair_ids), dtype=np.float32) * 10
        for mode in (2, 4):
            want, want_i = fo.score(as_f32, doc_off, doc_rows, np.arange(n), queries, pair_off, pair_ids, mode, sparse, 0.3)
            monkeypatch.delenv("FFX_DEBUG_VARIANT", raising=False)
            plan.set_pairs(pair_off, pair_ids, sparse, mode)
            for _ in range(2):  # (the second run starts from the tables the first one made)
                plan.run(0.3, exact_sum=True, sort=True)
                ff, interp, order = plan.fetch(interp=True, order=True)
                assert plan.stats()["kernel_variant"] == (311 if d == 768 else 312) + half
                assert np.array_equal(bits(ff), bits(want)) and np.array_equal(bits(interp), bits(want_i)), (mode, trial)
                assert np.array_equal(order.astype(np.int64), fo.order(want_i, pair_off))
            monkeypatch.setenv("FFX_DEBUG_VARIANT", "no_derive")
            plan.run(0.3, exact_sum=True, sort=True)
            ff2, interp2, _ = plan.fetch(interp=True)
            assert plan.stats()["kernel_variant"] == (111 if d == 768 else 112) + half
            assert np.array_equal(bits(ff2), bits(ff)) and np.array_equal(bits(interp2), bits(interp))
            monkeypatch.delenv("FFX_DEBUG_VARIANT", raising=False)
            # the fast summation order derives its rows too (511 / 512): within tolerance of the oracle, and the bits of its own
            # row-list form (211 / 212: the row list is made on the run that asks for it)
            plan.run(0.3, exact_sum=False)
            ff3, _, _ = plan.fetch()
            assert plan.stats()["kernel_variant"] == (511 if d == 768 else 512) + half
            assert np.abs(ff3 - want).max() <= FAST_TOL * 10
            monkeypatch.setenv("FFX_DEBUG_VARIANT", "no_derive")
            plan.run(0.3, exact_sum=False)
            ff3_list, _, _ = plan.fetch()
            assert plan.stats()["kernel_variant"] == (211 if d == 768 else 212) + half and np.array_equal(bits(ff3_list), bits(ff3))
            monkeypatch.delenv("FFX_DEBUG_VARIANT", raising=False)
            # ... and through ffx_score, slices of whole queries
            monkeypatch.setenv("FFX_DEBUG_VARIANT", "no_small_path,score_slice_pairs=250")
            ff4, interp4, order4, st = native.score(idx, queries, pair_off, pair_ids, mode, sparse, 0.3, sort=True)
            assert st["kernel_variant"] == (311 if d == 768 else 312) + half
            assert np.array_equal(bits(ff4), bits(want)) and np.array_equal(bits(interp4), bits(want_i)) and np.array_equal(order4, order)
        # one row per candidate (FIRSTP, PASSAGE with passages that are NOT their own rows): the row is looked up inside gather_dot
        # as well (variants 411 / 412): no expand launch, no row list; the row-list form (no_derive) gives the same bits
        psg_row = rng.permutation(n).astype(np.int32)
        idx.set_psg_rows(psg_row)
        for mode, pool in ((3, n_docs), (1, n)):
            ids1 = np.concatenate([rng.choice(pool, size=k, replace=False) for k in lens] + [np.zeros(0, dtype=np.int64)]).astype(np.int32)
            want, want_i = fo.score(as_f32, doc_off, doc_rows, psg_row, queries, pair_off, ids1, mode, sparse, 0.3)
            monkeypatch.delenv("FFX_DEBUG_VARIANT", raising=False)
            plan.set_pairs(pair_off, ids1, sparse, mode)
            plan.run(0.3, exact_sum=True, sort=True)
            ff, interp, order = plan.fetch(interp=True, order=True)
            assert plan.stats()["kernel_variant"] == (411 if d == 768 else 412) + half
            assert np.array_equal(bits(ff), bits(want)) and np.array_equal(bits(interp), bits(want_i)), (mode, trial)
            assert np.array_equal(order.astype(np.int64), fo.order(want_i, pair_off))
            monkeypatch.setenv("FFX_DEBUG_VARIANT", "no_derive")
            plan.run(0.3, exact_sum=True, sort=True)
            ff2, interp2, _ = plan.fetch(interp=True)
            assert plan.stats()["kernel_variant"] == (111 if d == 768 else 112) + half
            assert np.array_equal(bits(ff2), bits(ff)) and np.array_equal(bits(interp2), bits(interp))
            plan.run(0.3, exact_sum=False)  # (the fast order from the row list: 211 / 212 ...)
            fast_list, _, _ = plan.fetch()
            assert plan.stats()["kernel_variant"] == (211 if d == 768 else 212) + half
            monkeypatch.delenv("FFX_DEBUG_VARIANT", raising=False)
            plan.run(0.3, exact_sum=False)  # (... and with the row looked up in the kernel: 611 / 612, same bits)
            fast, _, _ = plan.fetch()
            assert plan.stats()["kernel_variant"] == (611 if d == 768 else 612) + half and np.array_equal(bits(fast), bits(fast_list))
            assert np.abs(fast - want).max() <= FAST_TOL * 10
            monkeypatch.setenv("FFX_DEBUG_VARIANT", "no_small_path,score_slice_pairs=250")
            ff4, interp4, order4, st = native.score(idx, queries, pair_off, ids1, mode, sparse, 0.3, sort=True)
            assert st["kernel_variant"] == (411 if d == 768 else 412) + half
            assert np.array_equal(bits(ff4), bits(want)) and np.array_equal(bits(interp4), bits(want_i)) and np.array_equal(order4, order)
            monkeypatch.delenv("FFX_DEBUG_VARIANT", raising=False)
            ff5, interp5, order5, st = native.score(idx, queries, pair_off, ids1, mode, sparse, 0.3, sort=True)  # (small batch: one submission)
            assert np.array_equal(bits(ff5), bits(want)) and np.array_equal(bits(interp5), bits(want_i)) and np.array_equal(order5, order)
        idx.set_psg_rows(np.arange(n))
        lens = lens[::-1]
    plan.close()
    idx.close()


@pytest.mark.parametrize("store", [np.float32, np.float16])
@pytest.mark.parametrize("d", [64, 128, 256, 384, 512, 768, 1024, 8, 24, 1000, 4100,
                               1536, 2048, 3072, 4096, 6144, 8192])  # second line: S x (768 | 1024), scored as S short rows
def test_every_kernel_variant_exact(d, store):
    """Each gather_dot instantiation (and the generic kernel), float32 and float16 rows, against
    the oracle and against NumPy itself, bit for bit."""
    n, nq, k = 257, 3, 150
    vectors = synth.rows(31 + d, 1.0, 0, n, d).astype(store)
    queries = synth.rows(32 + d, 1.0, 0, nq, d)
    idx = native.DeviceIndex(n, d, dtype=store)
    idx.upload(vectors, 0)
    idx.set_doc_rows(np.arange(n + 1), np.arange(n))
    idx.set_psg_rows(np.arange(n))
    rng = np.random.default_rng(d)
    pair_off = np.arange(nq + 1) * k
    pair_ids = np.concatenate([rng.choice(n, size=k, replace=False) for _ in range(nq)]).astype(np.int32)
    ff, _, _, st = native.score(idx, queries, pair_off, pair_ids, 2, exact_sum=True)
    want, _ = fo.score(vectors, np.arange(n + 1), np.arange(n), np.arange(n), queries, pair_off, pair_ids, 2)
    assert np.array_equal(bits(ff), bits(want)), f"d={d} variant {st['kernel_variant']}"
    assert (st["kernel_variant"] % 10000 >= 1000) == (store == np.float16)
    rows_found_in_kernel = {768: 411, 1024: 412, 512: 413, 384: 414, 256: 415, 128: 416, 64: 417}
    if d in rows_found_in_kernel:  # one row per candidate: no row list; the row-list form (no_derive) gives the same bits
        assert st["kernel_variant"] % 1000 == rows_found_in_kernel[d]
        os.environ["FFX_DEBUG_VARIANT"] = "no_derive"
        try:
            ff_list, _, _, st_list = native.score(idx, queries, pair_off, pair_ids, 2, exact_sum=True)
        finally:
            del os.environ["FFX_DEBUG_VARIANT"]
        assert st_list["kernel_variant"] % 1000 == rows_found_in_kernel[d] - 300 and np.array_equal(bits(ff_list), bits(ff))
    if d in (1536, 2048, 3072, 4096, 6144, 8192):  # S * 10000 + the variant of the short-row kernel
        base = 768 if d % 768 == 0 else 1024
        assert st["kernel_variant"] // 10000 == d // base and st["kernel_variant"] % 1000 == (111 if base == 768 else 112)
    ref = np.sum(queries[np.repeat(np.arange(nq), k)] * vectors[pair_ids], axis=1)  # NumPy itself
    assert ref.dtype == np.float32
    assert np.array_equal(bits(ff), bits(ref))
    ff_fast, _, _, st = native.score(idx, queries, pair_off, pair_ids, 2, exact_sum=False)
    assert np.abs(ff_fast - want).max() <= FAST_TOL * max(1.0, float(np.abs(want).max())), f"d={d} variant {st['kernel_variant']}"
    fast_rows_found_in_kernel = {768: 611, 1024: 612, 512: 613, 256: 614, 384: 414, 128: 416, 64: 417}  # (384 / 128 / 64: the exact kernels serve)
    if d in fast_rows_found_in_kernel:
        assert st["kernel_variant"] % 1000 == fast_rows_found_in_kernel[d]
        os.environ["FFX_DEBUG_VARIANT"] = "no_derive"
        try:
            fast_list, _, _, st_list = native.score(idx, queries, pair_off, pair_ids, 2, exact_sum=False)
        finally:
            del os.environ["FFX_DEBUG_VARIANT"]
        assert st_list["kernel_variant"] % 1000 == fast_rows_found_in_kernel[d] - (400 if d in (768, 1024, 512, 256) else 300)
        assert np.array_equal(bits(fast_list), bits(ff_fast))
    idx.close()


@pytest.mark.parametrize("store", [np.float32, np.float16])
def test_any_row_length_up_to_eight_leaves_is_bit_exact(store):
    """gather_dot_exact_anyd_kernel (run-time leaf table, every lane loads what it sums): row lengths with
    1, 2, 3, 4 .. 8 leaves of NumPy's tree, with and without trailing elements (len % 8), balanced and
    unbalanced recursions (d = 260 gives leaves 128 | 64 + 68) -- against NumPy itself, bit for bit;
    beyond 8 leaves (d > 1024) and below 8 elements the staged generic kernel still answers."""
    rng = np.random.default_rng(77)
    dims = [8, 9, 12, 15, 16, 17, 20, 31, 36, 100, 124, 127, 129, 130, 132, 136, 200, 255, 257, 260, 300, 383, 385, 388, 500, 513, 516, 640, 700,
            767, 769, 772, 900, 1000, 1016, 1020, 1023, 5, 1032, 1100]
    for d in dims:
        n, nq, k = 140, 3, 90
        vectors = (rng.standard_normal((n, d)) * 1.5).astype(store)
        queries = rng.standard_normal((nq, d)).astype(np.float32)
        idx = native.DeviceIndex(n, d, dtype=store)
        idx.upload(vectors, 0)
        idx.set_doc_rows(np.arange(n + 1), np.arange(n))
        idx.set_psg_rows(np.arange(n))
        pair_off = np.array([0, 1, k + 1, 2 * k + 1 - 7], dtype=np.int64)  # ragged, so tiles end inside a query
        pair_ids = rng.integers(0, n, size=int(pair_off[-1])).astype(np.int32)
        ff, _, _, st = native.score(idx, queries, pair_off, pair_ids, 1, exact_sum=True)
        ref = np.sum(queries[np.repeat(np.arange(nq), np.diff(pair_off))] * vectors[pair_ids], axis=1)
        assert ref.dtype == np.float32
        assert np.array_equal(bits(ff), bits(ref)), f"d={d} variant {st['kernel_variant']}"
        variant = st["kernel_variant"] % 1000

        def leaves(n):  # NumPy's pairwise recursion
            if n <= 128:
                return [n]
            n2 = n // 2 - (n // 2) % 8
            return leaves(n2) + leaves(n - n2)

        shape = leaves(d)
        if len(shape) <= 8 and min(shape) >= 8:
            assert variant == 130 + len(shape), (d, variant, shape)
        elif d < 8:
            assert variant == 198, (d, variant, shape)  # the staged generic kernel
        else:  # more than 8 leaves: segments of <= 8 leaves through the same kernel (next test)
            assert st["kernel_variant"] % 1000 == 140 and 2 <= st["kernel_variant"] // 10000 <= 64, (d, st["kernel_variant"])
        idx.close()


@pytest.mark.parametrize("store", [np.float32, np.float16])
def test_long_rows_of_any_length_are_bit_exact_through_segments(store):
    """Rows of more than 8 NumPy leaves that are not S x a specialised length (d = 1100, 1280, 2000, 5000 ...): cut along
    NumPy's pairwise recursion into segments of at most 8 leaves each -- unbalanced cuts included (1100 -> 544 | 556) --,
    every segment summed by the any-d kernel, the segment sums added in the order of the recursion.  Bit for bit
    against NumPy itself for 26 lengths in (1024, 8192]; the staged generic kernel no longer runs for them."""
    rng = np.random.default_rng(123)
    dims = [1025, 1032, 1100, 1280, 1281, 1500, 1537, 1999, 2000, 2047, 2049, 2500, 2560, 3000, 3073, 3500, 4000, 4100, 4097, 5000, 5555,
            6000, 7000, 7777, 8000, 8191]

    def leaves(n):
        if n <= 128:
            return [n]
        n2 = n // 2 - (n // 2) % 8
        return leaves(n2) + leaves(n - n2)

    def pieces(n):  # the segments: follow the recursion until a piece has <= 8 leaves
        if len(leaves(n)) <= 8:
            return [n]
        n2 = n // 2 - (n // 2) % 8
        return pieces(n2) + pieces(n - n2)

    for d in dims:
        n, nq, k = 70, 3, 60
        vectors = (rng.standard_normal((n, d)) * 1.5).astype(store)
        queries = rng.standard_normal((nq, d)).astype(np.float32)
        idx = native.DeviceIndex(n, d, dtype=store)
        idx.upload(vectors, 0)
        idx.set_doc_rows(np.arange(n + 1), np.arange(n))
        idx.set_psg_rows(np.arange(n))
        pair_off = np.array([0, 1, k + 1, 2 * k + 1 - 7], dtype=np.int64)
        pair_ids = rng.integers(0, n, size=int(pair_off[-1])).astype(np.int32)
        ff, _, _, st = native.score(idx, queries, pair_off, pair_ids, 1, exact_sum=True)
        ref = np.sum(queries[np.repeat(np.arange(nq), np.diff(pair_off))] * vectors[pair_ids].astype(np.float32), axis=1)
        assert ref.dtype == np.float32
        assert np.array_equal(bits(ff), bits(ref)), f"d={d} variant {st['kernel_variant']}"
        assert st["kernel_variant"] % 1000 == 140 and st["kernel_variant"] // 10000 == len(pieces(d)), (d, st["kernel_variant"], pieces(d))
        assert (st["kernel_variant"] % 10000 >= 1000) == (store == np.float16)
        idx.close()


@pytest.mark.parametrize("store", [np.float32, np.float16])
def test_synthetic_fill_matches_host_generator(store):
    d, n = 768, 1000
    idx = native.DeviceIndex(n, d, dtype=store)
    idx.fill_synthetic(seed=5, scale=0.0625, row_offset=0, n_rows=n, first_global_row=123456)
    rows = np.arange(0, n, dtype=np.int32)  # every element: narrowing to half must round twice, as NumPy does
    got = idx.gather(rows)
    assert got.dtype == store
    want = synth.rows(5, 0.0625, 123456, n, d).astype(store)
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))
    idx.fill_synthetic(seed=6, scale=float(1 / np.sqrt(768)), row_offset=0, n_rows=n, first_global_row=7)
    want = synth.rows(6, float(1 / np.sqrt(768)), 7, n, d).astype(store)
    assert np.array_equal(idx.gather(rows).view(np.uint8), want.view(np.uint8))
    idx.close()


def test_full_size_properties():
    """Properties that hold at any size, checked on a few hundred MB: every candidate is a
    distinct synthetic row, so (a) scoring a query against its own row list twice gives the
    same bits, (b) scores are invariant to how candidates are split into queries, (c) a
    sampled subset matches the oracle bit for bit."""
    d, n, nq, k = 768, 200_000, 64, 2000
    idx = native.DeviceIndex(n, d)
    idx.fill_synthetic(seed=9, scale=0.0625, row_offset=0, n_rows=n)
    idx.set_doc_rows(np.arange(n + 1), np.arange(n))
    idx.set_psg_rows(np.arange(n))
    queries = synth.rows(10, 0.0625, 0, nq, d)
    pair_ids = (synth.uniform_u32(np.arange(nq * k, dtype=np.uint64), 4) % np.uint32(n)).astype(np.int32)
    pair_off = np.arange(nq + 1, dtype=np.int64) * k
    a, _, _, _ = native.score(idx, queries, pair_off, pair_ids, 2)
    b, _, _, _ = native.score(idx, queries, pair_off, pair_ids, 1)
    assert np.array_equal(bits(a), bits(b))
    # split every query in two halves (same query vector twice): same scores
    q2 = np.repeat(queries, 2, axis=0)
    off2 = np.arange(2 * nq + 1, dtype=np.int64) * (k // 2)
    c, _, _, _ = native.score(idx, q2, off2, pair_ids, 2)
    assert np.array_equal(bits(a), bits(c))
    # oracle spot check on 2000 sampled pairs
    sel = np.random.default_rng(0).choice(nq * k, size=2000, replace=False)
    rows = np.concatenate([synth.rows(9, 0.0625, int(r), 1, d) for r in pair_ids[sel]])  # re-derived on the host
    want = fo.row_scores(queries, rows, (sel // k).astype(np.int32), np.arange(sel.shape[0], dtype=np.int32))
    assert np.array_equal(bits(a[sel]), bits(want))
    idx.close()



def test_sharded_scorer_single_rank_nccl():
    """The multi-GPU code path end to end on one GPU: a 1-rank RCCL communicator created through the
    C ABI (ffx_comm_init), plans on the compute stream, the exchange on plan-owned memory
    (zero-copy), the owner-side merge kernel.  Own interpreter: the worker asserts that the whole
    path runs without PyTorch in the process."""
    import subprocess
    import sys
    from pathlib import Path

    worker = Path(__file__).resolve().parent / "_nccl1_worker.py"
    out = subprocess.run([sys.executable, str(worker)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "sharded 1-rank ok" in out.stdout


# (world 5: the most PROCESSES one GPU box lets share its card -- its process guard allows six on the GPU at a time, and this
# pytest process is one of them.  bench.py is one rank per process, so five it is here; the eight-rank rendezvous, piece
# tables and owner routing run against this device with two ranks per process in
# test_owner_routed_fresh_batches_with_ranks_sharing_the_gpu[8], test_index_api.py and test_ondisk.py, and without a device in
# tests/test_sharded_native.py)
@pytest.mark.parametrize("world,sharding,spelling", [(2, "rows", "--transport host"), (3, "rows", "--transport host"), (2, "queries", "--transport host"),
                                                     (2, "rows", "--dist-backend gloo"), (5, "rows", "--transport host")])
def test_sharded_bench_leg_with_ranks_sharing_the_gpu(world, sharding, spelling):
    """`python bench.py --gpus N` exactly as the driver starts it (no outer launcher: bench.py starts
    its own ranks): rows sharded by document, two plans alternating between a compute and a
    communication stream, owner-side merge, with `world` ranks on this one GPU; the exchange goes
    through the shared-memory transport because RCCL refuses ranks that share a device.  Rank 0's
    scores are checked against the oracle inside bench.py (`parity_on_cpu_sample`)."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", str(world), *spelling.split(),  # (--dist-backend gloo: round 1's spelling of the same)
           "--placement-tries", "0", "--rows", "300000", "--queries", "90", "--k", "700", "--steps", "4", "--warmup", "1", "--cpu-sample-queries", "12",
           "--cpu-seconds", "0.2", "--sharding", sharding]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = next(ln for ln in out.stdout.splitlines() if ln.startswith("{"))
    result = json.loads(line)
    assert result["n_gpus"] == world and result["parity_on_cpu_sample"] == "bit-exact"
    assert sum(result["roofline"]["gathered_rows_per_rank"]) == 90 * 700
    if sharding == "rows":
        # every rank takes in ITS block of a fresh batch: offsets + (ordinal, first-stage score) per pair + its query vectors
        assert sum(result["stage_h2d_bytes_per_rank"]) == (90 + world) * 8 + 90 * 700 * 8 + 90 * 768 * 4
        assert max(result["stage_h2d_bytes_per_rank"]) < 1.2 * sum(result["stage_h2d_bytes_per_rank"]) / world
        assert result["fresh_batch_step_ms"] > 0 and result["host_to_host_batch_ms"] > 0
        assert result["transport"] == "host" and len(result["exchange_bytes_per_rank"]["sent"]) == world
        assert sum(result["owned_pairs_per_rank"]) == 90 * 700
        assert sum(result["exchange_bytes_per_rank"]["sent"]) == sum(result["exchange_bytes_per_rank"]["received"])
        # ranks that share a GPU over shared memory are a rehearsal: the line must not read as a scaling point
        assert result["scaling"] == "invalid: host transport" and result["value_is_rccl"] is False


def test_sharded_bench_falls_back_together_when_rccl_cannot_come_up():
    """The default transport on a node where RCCL cannot be used (here: two ranks masked onto the same GPU,
    which RCCL refuses): every rank probes RCCL under a watchdog, the ranks vote over the shared-memory side
    channel and ALL of them continue on the shared-memory transport; the line says so, the run is correct and
    exits 0 -- a scaling run neither hangs on nor dies of a fabric that will not come up."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "2", "--placement-tries", "0", "--rows", "300000", "--queries", "90", "--k", "700", "--steps", "3",
           "--warmup", "1", "--cpu-sample-queries", "12", "--cpu-seconds", "0.2"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = "0"
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = next(ln for ln in out.stdout.splitlines() if ln.startswith("{"))
    result = json.loads(line)
    assert result["n_gpus"] == 2 and result["parity_on_cpu_sample"] == "bit-exact"
    assert result["transport"].startswith("host (rccl unavailable: rank "), result["transport"]
    assert result["rccl_ranks"] == 0
    assert "fall" in out.stderr and "shared-memory transport" in out.stderr
    # ... and the line cannot be mistaken for a measurement of the xGMI path
    assert result["scaling"] == "invalid: host transport" and result["value_is_rccl"] is False
    # --require-rccl: no fallback, every rank leaves with a non-zero status and no JSON line is printed
    out = subprocess.run(cmd + ["--require-rccl"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode != 0
    assert not any(ln.startswith("{") for ln in out.stdout.splitlines())
    assert "--require-rccl" in out.stderr


def test_rehearsal_of_one_rank_of_eight_on_one_gpu():
    """`bench.py --as-rank-of 8 --as-rank 5`: one process stands in for rank 5 of 8 (its eighth of the rows, the candidates
    it holds, the block of queries it owns; the exchange moves the same bytes through this GPU's memory).  The pairs the
    rank holds are checked against the oracle inside bench.py; the line is marked as a rehearsal."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    cmd = [sys.executable, str(root / "bench.py"), "--as-rank-of", "8", "--as-rank", "5", "--placement-tries", "0", "--rows", "400000", "--queries", "160", "--k", "900",
           "--steps", "6", "--warmup", "2", "--cpu-sample-queries", "6", "--cpu-seconds", "0.2"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    result = json.loads(next(ln for ln in out.stdout.splitlines() if ln.startswith("{")))
    assert result["n_gpus"] == 1 and result["scaling"].startswith("invalid: single-GPU rehearsal") and result["value_is_rccl"] is False
    reh = result["rehearsal"]
    assert (reh["as_rank"], reh["of"]) == (5, 8) and reh["owned_queries"] == 20 and reh["owned_pairs"] == 20 * 900
    assert abs(reh["local_pairs"] - 160 * 900 / 8) < 0.1 * 160 * 900 / 8
    assert result["parity_on_cpu_sample"].startswith("bit-exact on the ")
    # the fresh-batch loop of the rehearsed rank: its own block goes up (1/8 of the batch), the peers' messages come out of the inbox
    assert result["stage_h2d_bytes_per_rank"][5] == 21 * 8 + 20 * 900 * 8 + 20 * 768 * 4
    assert reh["fresh_batch_step_ms"] > 0


def test_device_routing_equals_host_routing(monkeypatch):
    """The sharded scorer routes a batch on the device when the shards are document ranges (one upload of the ids,
    then mark / scan / compact / stable radix pass by holder in HBM).  Against the host routing (FFX_DEBUG_VARIANT=route_host)
    on one rank: same dense, interpolated scores and order, bit for bit, for ragged batches (empty queries, one
    candidate, several passages per document, every mode); the same NotFound for an ordinal outside every shard and
    for one inside the range that has no rows."""
    rng = np.random.default_rng(21)
    d, n_docs = 96, 700
    per_doc = rng.integers(0, 4, size=n_docs)  # some documents hold no rows at all
    per_doc[:5] = 1
    doc_off = np.concatenate([[0], np.cumsum(per_doc)]).astype(np.int64)
    n = int(doc_off[-1])
    vectors = synth.rows(31, 0.1, 0, n, d)
    queries = synth.rows(32, 0.1, 0, 7, d)
    shard = native.DeviceIndex(n, d)
    shard.upload(vectors, 0)
    shard.set_doc_rows(doc_off, rng.permutation(n).astype(np.int32))
    shard.set_psg_rows(np.arange(n, dtype=np.int32))
    with_rows = np.flatnonzero(per_doc > 0)
    lens = [0, 1, 300, 0, 77, 256, 5]
    pair_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pair_ids = np.concatenate([rng.choice(with_rows, size=k, replace=False) for k in lens]).astype(np.int32)
    sparse = (rng.random(len(pair_ids)) * 9).astype(np.float32)
    comm = native.Comm(native.COMM_HOST, 0, 0, 1, native.Comm.unique_id(native.COMM_HOST))
    results = {}
    for route in ("device", "host"):
        if route == "host":
            monkeypatch.setenv("FFX_DEBUG_VARIANT", "route_host")
        else:
            monkeypatch.delenv("FFX_DEBUG_VARIANT", raising=False)
        sc = native.Sharded(shard, comm, np.array([0, n_docs], dtype=np.int64))
        sc.set_queries(queries)
        for mode in (2, 3, 4):
            for sp in (sparse, None):
                sc.set_pairs(pair_off, pair_ids, sp, mode)
                sc.run(0.3, sort=True)
                sc.sync()
                results[(route, mode, sp is None)] = sc.fetch(ff=True, interp=sp is not None, order=True)
        for value, at in ((n_docs, 40), (-1, 3), (int(np.flatnonzero(per_doc == 0)[0]), 100)):
            bad = pair_ids.copy()
            bad[at] = value
            with pytest.raises(native.NotFound) as e:
                sc.set_pairs(pair_off, bad, sparse, 2)
            assert e.value.pair == at, (route, value)
        sc.set_pairs(pair_off, pair_ids, sparse, 2)  # still usable
        sc.run(0.3, sort=True)
        sc.sync()
        again = sc.fetch(ff=True, interp=True, order=True)
        for x, y in zip(again, results[(route, 2, False)]):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
        sc.close()
    for mode in (2, 3, 4):
        for no_sparse in (False, True):
            a, b = results[("device", mode, no_sparse)], results[("host", mode, no_sparse)]
            for x, y in zip(a, b):
                assert (x is None) == (y is None)
                if x is not None:
                    assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    comm.close()
    shard.close()


@pytest.mark.parametrize("world", [1, 2, 3, 8])  # (five processes of one rank ran here until eight ranks in four processes did: 28 s the suite can do without)
def test_owner_routed_fresh_batches_with_ranks_sharing_the_gpu(world):
    """ffx_sharded_stage_owned / ffx_sharded_score: every rank is handed only the queries it owns, finds the holders on the
    device and sends them the ordinals in a first all-to-all; results equal the single-GPU scorer over the whole index bit for
    bit -- sliced and unsliced, document ranges and explicit holders, every mode, several different batches in flight, and
    an ordinal without rows is the same IndexError on every rank (tests/_sharded_owned_worker.py)."""
    from pathlib import Path

    from test_sharded_native import run_ranks

    # (the box admits six processes to its GPU, this one included: eight ranks are four processes of two ranks, a thread each)
    per_process = 2 if world > 5 else 1
    codes, outs = run_ranks(world, Path(__file__).resolve().parent / "_sharded_owned_worker.py", timeout=600, ranks_per_process=per_process)
    assert codes == [0] * len(codes) and len(codes) == -(-world // per_process), "\n".join(o[0][-1500:] + o[1][-3000:] for o in outs)
    assert f"owner-routed world={world}:" in outs[0][0] and "checked bit-exact" in outs[0][0]


def test_sharded_bench_leg_also_runs_under_torchrun():
    """The other way the driver may start N ranks: `python -m torch.distributed.run ... bench.py --gpus N`
    (RANK / WORLD_SIZE from the launcher; the ranks find each other through the launcher's run id)."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29712", str(root / "bench.py"), "--gpus", "2", "--transport", "host",
           "--placement-tries", "0", "--rows", "200000", "--queries", "60", "--k", "500", "--steps", "3", "--warmup", "1", "--cpu-sample-queries", "8", "--cpu-seconds", "0.2"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    result = json.loads(next(ln for ln in out.stdout.splitlines() if ln.startswith("{")))
    assert result["n_gpus"] == 2 and result["parity_on_cpu_sample"] == "bit-exact"


def test_single_rank_rccl_bench_leg_matches_the_plain_leg():
    """`--force-sharded` with one rank runs the whole sharded machinery (RCCL communicator, exchange =
    a device copy, owner merge) on the N=1 workload: same scores, and a step time close to the plain
    leg's (the sharded path adds one copy and one merge kernel)."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    common = ["--placement-tries", "0", "--rows", "2000000", "--queries", "400", "--k", "5000", "--steps", "10", "--warmup", "3", "--cpu-sample-queries", "4",
              "--cpu-seconds", "0.2", "--no-regions", "--no-traffic"]
    vals = {}
    for name, extra in (("plain", []), ("sharded", ["--force-sharded"])):
        out = subprocess.run([sys.executable, str(root / "bench.py"), *common, *extra], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
        vals[name] = json.loads(next(ln for ln in out.stdout.splitlines() if ln.startswith("{")))
        assert vals[name]["parity_on_cpu_sample"] == "bit-exact"
    assert vals["sharded"]["rccl_ranks"] == 1
    assert vals["sharded"]["value"] > 0.9 * vals["plain"]["value"], (vals["sharded"]["value"], vals["plain"]["value"])
    # documents of several passages: the sharded scorer's plans derive their rows inside gather_dot too (variant 311), replayed
    # and fresh batches bit-exact against the oracle
    out = subprocess.run([sys.executable, str(root / "bench.py"), "--layout", "multi", "--force-sharded", "--placement-tries", "0", "--rows", "1000000", "--queries", "300",
                          "--k", "1500", "--steps", "4", "--warmup", "1", "--cpu-sample-queries", "8", "--cpu-seconds", "0.2"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    multi = json.loads(next(ln for ln in out.stdout.splitlines() if ln.startswith("{")))
    assert multi["parity_on_cpu_sample"] == "bit-exact" and multi["fresh_batch_parity_on_cpu_sample"] == "bit-exact"
    assert "variant 311" in multi["roofline"]["kernel"], multi["roofline"]["kernel"]


def test_bench_measures_hbm_traffic_with_pmc_passes_of_its_own():
    """The 1-GPU bench line's roofline.traffic is measured in the run itself: bench.py starts rocprofv3 --pmc
    child passes (FETCH_SIZE, WRITE_SIZE; one counter per pass, no tracing) of the same workload before it
    touches the GPU.  On 2 M x 768 fp32 rows (6 GB, far beyond the 256 MB Infinity Cache) with 2 M candidates the
    kernel must fetch what the algorithm needs and little more."""
    import json
    import shutil
    import subprocess
    import sys
    from pathlib import Path

    if shutil.which("rocprofv3") is None and not os.path.exists("/opt/rocm/bin/rocprofv3"):
        pytest.skip("rocprofv3 is not installed")
    root = Path(__file__).resolve().parent.parent
    cmd = [sys.executable, str(root / "bench.py"), "--rows", "2000000", "--queries", "400", "--k", "5000", "--steps", "5", "--warmup", "2",
           "--cpu-sample-queries", "0", "--no-regions"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    roof = json.loads(next(ln for ln in out.stdout.splitlines() if ln.startswith("{")))["roofline"]
    if roof["traffic"] is None:  # the bench line still came out; counters are not readable for this user / on this box
        pytest.skip(f"PMC passes unavailable here: {roof['traffic_source']}")
    assert roof["traffic_source"].startswith("measured in this run")
    assert 0.9 < roof["traffic_over_algorithmic"] < 1.25, roof
    assert roof["traffic_write_bytes"] < 0.01 * roof["traffic_fetch_bytes"]


# ---- product-quantized rows (decode-then-dot and ADC) ------------------------------------------------
from _cases import PQ_CASES, load_pq_case  # noqa: E402


@pytest.mark.parametrize("name", PQ_CASES)
def test_pq_decode_then_dot_bit_identical_to_reference(name):
    """uint8 codes in HBM + codebook in L2, scored by the decode-then-dot kernels, against the
    frames the reference produced with the same codewords (index/base.py:292-293)."""
    case = load_pq_case(name)
    m = case.codes.shape[1]
    idx = native.DeviceIndex(case.codes.shape[0], m, dtype=np.uint8)
    idx.set_codebook(case.codewords)
    idx.upload(case.codes, 0)
    idx.set_doc_rows(case.doc_off, case.doc_rows)
    idx.set_psg_rows(case.psg_row)
    assert np.array_equal(idx.gather(np.arange(5, dtype=np.int32)), case.codes[:5])
    for mode in case.modes():
        pair_off, pair_ids, sparse, names = case.pairs(mode)
        ff, interp, order, st = native.score(idx, case.queries, pair_off, pair_ids, MODE_NO[mode], sparse, 0.1, exact_sum=True, sort=True)
        pair_q = np.repeat(np.arange(len(case.q_names)), np.diff(pair_off))
        for key, got in ((f"{mode}_ff", ff), (f"{mode}_int0.1", interp)):
            exp = case.expected(key)
            want = np.array([exp[(case.q_names[q], n)] for q, n in zip(pair_q, names)], dtype=np.float32)
            assert np.array_equal(bits(got), bits(want)), (name, key, st["kernel_variant"])
        # subspaces == NumPy leaves (d=768, M=8): exact ADC; otherwise decode-then-dot (LDS / generic)
        # (single-row candidates -- PASSAGE / FIRSTP here -- take the fused pairs kernel 24xx, multi-passage modes the row-list form 22xx)
        exact_adc = 2408 if mode in ("PASSAGE", "FIRSTP") else 2208
        assert st["kernel_variant"] == {"pq_d768_m8": exact_adc, "pq_d768_m16": 2116, "pq_d1024_m32": 2132, "pq_d60_m5": 2199}[name]
        assert np.array_equal(order.astype(np.int64), fo.order(interp, pair_off))
        if name == "pq_d768_m8":  # the decode-then-dot kernel must give the same bits
            import os

            os.environ["FFX_DEBUG_VARIANT"] = "pq_no_adc"
            try:
                ff2, _, _, st2 = native.score(idx, case.queries, pair_off, pair_ids, MODE_NO[mode], exact_sum=True)
            finally:
                del os.environ["FFX_DEBUG_VARIANT"]
            assert st2["kernel_variant"] == 2108 and np.array_equal(bits(ff2), bits(ff))
        # ADC (fast order): same value up to fp32 rounding of a different summation order
        fast, _, _, st = native.score(idx, case.queries, pair_off, pair_ids, MODE_NO[mode], exact_sum=False)
        assert st["kernel_variant"] == (exact_adc if name == "pq_d768_m8" else 2300)  # the exact table is the faster one where it applies
        assert np.abs(fast - ff).max() <= FAST_TOL * max(1.0, float(np.abs(ff).max()))
    idx.close()


def test_pq_synthetic_codes_fill():
    n, m, ks = 5000, 8, 256
    idx = native.DeviceIndex(n, m, dtype=np.uint8)
    idx.set_codebook(synth.rows(3, 1.0, 0, m * ks, 96).reshape(m, ks, 96))
    idx.fill_synthetic(seed=7, scale=1.0, row_offset=0, n_rows=n, first_global_row=11)
    got = idx.gather(np.arange(n, dtype=np.int32))
    want = ((synth.mix64(np.arange(11 * m, (11 + n) * m, dtype=np.uint64), 7) >> np.uint64(32)) % np.uint64(ks)).astype(np.uint8).reshape(n, m)
    assert np.array_equal(got, want)
    idx.close()


def test_pq_codes_outside_the_codebook_are_refused_on_upload():
    """With fewer than 256 codewords per subspace, a code byte >= Ks (a corrupt or foreign index
    file) would index past the codebook and the ADC tables: uploads reject it, naming row and
    subspace (and the kernels clamp, so even a code that slipped through reads defined memory)."""
    m, ks, ds = 8, 16, 96
    idx = native.DeviceIndex(100, m, dtype=np.uint8)
    idx.set_codebook(synth.rows(3, 1.0, 0, m * ks, ds).reshape(m, ks, ds))
    codes = (np.arange(100 * m, dtype=np.int64).reshape(100, m) % ks).astype(np.uint8)
    idx.upload(codes, 0)
    bad = codes.copy()
    bad[37, 5] = ks  # first illegal value
    with pytest.raises(native.FFXError, match=r"row 37 holds a code >= Ks = 16 in subspace 5"):
        idx.upload(bad, 0)
    with pytest.raises(native.FFXError, match=r"row 17 "):  # device row: 10 + (37 - 30)
        idx.upload(bad[30:50], 10)
    idx.close()


@pytest.mark.parametrize("m,ks", [(96, 256), (192, 256)])
def test_pq_adc_tables_larger_than_the_default_lds_window(m, ks):
    """exact_sum=False on a quantizer whose per-query table M*Ks*4 exceeds 64 KB (NanoPQ(96, 256) on
    d = 768 needs 96 KB: opt-in LDS up to the CU's 160 KB) or even the whole LDS (192 KB: tables are
    read through L2 instead): same scores as decode + dot, within the tolerance."""
    d, n, nq, k = 768, 500, 3, 100
    ds = d // m
    rng = np.random.default_rng(m)
    codewords = rng.standard_normal((m, ks, ds)).astype(np.float32) * 0.1
    codes = rng.integers(0, ks, size=(n, m)).astype(np.uint8)
    idx = native.DeviceIndex(n, m, dtype=np.uint8)
    idx.set_codebook(codewords)
    idx.upload(codes, 0)
    idx.set_doc_rows(np.arange(n + 1), np.arange(n))
    idx.set_psg_rows(np.arange(n))
    queries = rng.standard_normal((nq, d)).astype(np.float32)
    pair_off = np.arange(nq + 1, dtype=np.int64) * k
    pair_ids = np.concatenate([rng.choice(n, size=k, replace=False) for _ in range(nq)]).astype(np.int32)
    ff, _, _, st = native.score(idx, queries, pair_off, pair_ids, MODE_NO["PASSAGE"], exact_sum=False)
    assert st["kernel_variant"] == (2300 if m * ks * 4 <= 160 * 1024 else 2301)
    decoded = np.concatenate([codewords[j][codes[:, j]] for j in range(m)], axis=1)
    want = (queries[np.repeat(np.arange(nq), k)].astype(np.float64) * decoded[pair_ids].astype(np.float64)).sum(1)
    assert np.abs(ff - want).max() <= FAST_TOL * max(1.0, np.abs(want).max())
    idx.close()


# ---- BASELINE.json's full size ----------------------------------------------------------------------------
def test_baseline_config_full_size_properties():
    """8 841 823 x 768 fp32 in HBM, 6980 queries x k=5000 (34.9 M pairs, the bench workload), checked
    through properties that need no 27 GB host copy:
      * the device order is a per-query permutation and scores are non-increasing along it
      * reversing every candidate list reverses the scores bit for bit (no dependence on tiling)
      * MAXP == FIRSTP == PASSAGE on one-passage documents
      * 4096 sampled pairs equal the oracle on rows re-derived on the host
    """
    import importlib.util
    from pathlib import Path

    spec = importlib.util.spec_from_file_location("bench", Path(__file__).resolve().parent.parent / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n, d, nq, k, seed = 8_841_823, 768, 6980, 5000, 2024
    scale = float(1 / np.sqrt(d))
    idx = native.DeviceIndex(n, d)
    idx.fill_synthetic(seed, scale, 0, n)
    idx.set_doc_rows(np.arange(n + 1, dtype=np.int64), np.arange(n, dtype=np.int32))
    idx.set_psg_rows(np.arange(n, dtype=np.int32))
    ids, sparse = bench.candidates(n, nq, k, seed)
    pair_off = np.arange(nq + 1, dtype=np.int64) * k
    queries = synth.rows(seed + 3, scale, 0, nq, d)
    plan = native.Plan(idx)
    plan.set_queries(queries)
    plan.set_pairs(pair_off, ids, sparse, 2)
    plan.run(0.1, exact_sum=True, sort=True)
    ff, interp, order = plan.fetch(interp=True, order=True)
    st = plan.stats()
    assert st["n_rows"] == nq * k and st["kernel_variant"] == 411  # (one row per candidate, found inside gather_dot: no row list)
    # order: permutation of each query's positions, scores non-increasing
    o2 = order.reshape(nq, k).astype(np.int64)
    assert np.array_equal(np.sort(o2, axis=1), pair_off[:-1, None] + np.arange(k)[None, :])
    along = interp[o2]
    assert (np.diff(along, axis=1) <= 0).all()
    ties = np.diff(along, axis=1) == 0
    assert (np.diff(o2, axis=1)[ties] > 0).all()  # exact ties keep first-stage order
    # reversed candidate lists
    rev = ids.reshape(nq, k)[:, ::-1].reshape(-1).copy()
    plan.set_pairs(pair_off, rev, sparse.reshape(nq, k)[:, ::-1].reshape(-1).copy(), 2)
    plan.run(0.1, exact_sum=True)
    ff_r, interp_r, _ = plan.fetch(interp=True)
    assert np.array_equal(bits(ff_r.reshape(nq, k)[:, ::-1]), bits(ff.reshape(nq, k)))
    assert np.array_equal(bits(interp_r.reshape(nq, k)[:, ::-1]), bits(interp.reshape(nq, k)))
    # modes agree on one-passage documents
    for mode in (1, 3):
        plan.set_pairs(pair_off, ids, None, mode)
        plan.run(0.0, exact_sum=True)
        assert np.array_equal(bits(plan.fetch()[0]), bits(ff)), mode
    # sampled pairs against the oracle
    sel = np.random.default_rng(1).choice(nq * k, size=4096, replace=False)
    rows = fo.synth_row_list(seed, scale, ids[sel], d)
    want = fo.row_scores(queries, rows, (sel // k).astype(np.int32), np.arange(sel.shape[0], dtype=np.int32))
    assert np.array_equal(bits(ff[sel]), bits(want))
    assert np.array_equal(bits(interp[sel]), bits(fo.interpolate(sparse[sel], want, 0.1)))
    plan.close()
    idx.close()


def _bench_module():
    import importlib.util
    from pathlib import Path

    spec = importlib.util.spec_from_file_location("bench", Path(__file__).resolve().parent.parent / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def _full_size_properties(idx, n_docs, nq, k, seed, scale, d, rows_of, store_label):
    """The size-independent checks of test_baseline_config_full_size_properties on an index of any store:
    order = per-query permutation with non-increasing scores, reversal invariance, MAXP == FIRSTP == PASSAGE on
    one-passage documents, 4096 sampled pairs against the oracle on rows re-derived on the host (``rows_of(ids)``)."""
    bench = _bench_module()
    ids, sparse = bench.candidates(n_docs, nq, k, seed)
    pair_off = np.arange(nq + 1, dtype=np.int64) * k
    queries = synth.rows(seed + 3, scale, 0, nq, d)
    plan = native.Plan(idx)
    plan.set_queries(queries)
    plan.set_pairs(pair_off, ids, sparse, 2)
    plan.run(0.1, exact_sum=True, sort=True)
    ff, interp, order = plan.fetch(interp=True, order=True)
    assert plan.stats()["n_rows"] == nq * k
    o2 = order.reshape(nq, k).astype(np.int64)
    assert np.array_equal(np.sort(o2, axis=1), pair_off[:-1, None] + np.arange(k)[None, :])
    along = interp[o2]
    assert (np.diff(along, axis=1) <= 0).all(), store_label
    ties = np.diff(along, axis=1) == 0
    assert (np.diff(o2, axis=1)[ties] > 0).all()
    rev = ids.reshape(nq, k)[:, ::-1].reshape(-1).copy()
    plan.set_pairs(pair_off, rev, sparse.reshape(nq, k)[:, ::-1].reshape(-1).copy(), 2)
    plan.run(0.1, exact_sum=True)
    ff_r, interp_r, _ = plan.fetch(interp=True)
    assert np.array_equal(bits(ff_r.reshape(nq, k)[:, ::-1]), bits(ff.reshape(nq, k))), store_label
    assert np.array_equal(bits(interp_r.reshape(nq, k)[:, ::-1]), bits(interp.reshape(nq, k)))
    for mode in (1, 3):
        plan.set_pairs(pair_off, ids, None, mode)
        plan.run(0.0, exact_sum=True)
        assert np.array_equal(bits(plan.fetch()[0]), bits(ff)), (store_label, mode)
    sel = np.random.default_rng(1).choice(nq * k, size=4096, replace=False)
    want = fo.row_scores(queries, rows_of(ids[sel]), (sel // k).astype(np.int32), np.arange(sel.shape[0], dtype=np.int32))
    assert np.array_equal(bits(ff[sel]), bits(want)), store_label
    assert np.array_equal(bits(interp[sel]), bits(fo.interpolate(sparse[sel], want, 0.1)))
    # the one-shot host-to-host call (sliced, pipelined) gives the same bits and the same order
    ff2, interp2, order2, _ = native.score(idx, queries, pair_off, ids, 2, sparse, 0.1, exact_sum=True, sort=True)
    assert np.array_equal(bits(ff2), bits(ff)) and np.array_equal(bits(interp2), bits(interp)) and np.array_equal(order2, order)
    plan.close()


def test_config2_full_size_at_k_1000():
    """BASELINE config 2 as written: 8 841 823 x 768 fp32 on one GPU, 6980 queries, k = 1000, Mode.MAXP (6.98 M pairs)."""
    n, d, seed = 8_841_823, 768, 2024
    scale = float(1 / np.sqrt(d))
    idx = native.DeviceIndex(n, d)
    idx.fill_synthetic(seed, scale, 0, n)
    idx.set_doc_rows(np.arange(n + 1, dtype=np.int64), np.arange(n, dtype=np.int32))
    idx.set_psg_rows(np.arange(n, dtype=np.int32))
    _full_size_properties(idx, n, 6980, 1000, seed, scale, d, lambda ids: fo.synth_row_list(seed, scale, ids, d), "fp32 k=1000")
    idx.close()


def test_config4_float16_rows_50m_full_size():
    """BASELINE config 4, first half: 50 M x 768 float16 rows = 76.8 GB resident on the one GPU (filled on the device), 6980
    queries x k = 5000.  The reference keeps the dtype it is given (index/memory.py:80-83) and NumPy multiplies float32
    queries by float16 rows in float32 (index/base.py:292-303): the oracle scores the rows rounded to float16 once."""
    n, d, seed = 50_000_000, 768, 2024
    scale = float(1 / np.sqrt(d))
    if native.device_info(0)["free_bytes"] < 90e9:
        pytest.skip("needs ~80 GB of free HBM")
    idx = native.DeviceIndex(n, d, dtype=np.float16)
    idx.fill_synthetic(seed, scale, 0, n)
    idx.set_doc_rows(np.arange(n + 1, dtype=np.int64), np.arange(n, dtype=np.int32))
    idx.set_psg_rows(np.arange(n, dtype=np.int32))
    _full_size_properties(idx, n, 6980, 5000, seed, scale, d,
                          lambda ids: fo.synth_row_list(seed, scale, ids, d).astype(np.float16).astype(np.float32), "fp16 50M")
    idx.close()


def test_config4_product_quantized_rows_50m_full_size():
    """BASELINE config 4, second half: 50 M rows of M = 8 product-quantizer codes (Ks = 256, Ds = 96), 6980 queries x
    k = 5000: the exact ADC path (fused pairs kernel) against decode + np.sum(q * d) (index/base.py:292-303,
    quantizer/nanopq.py:43-44) on 4096 sampled pairs whose codes are re-derived on the host."""
    n, d, m, ks, seed = 50_000_000, 768, 8, 256, 2024
    scale = float(1 / np.sqrt(d))
    book = synth.rows(seed + 9, scale, 0, m * ks, d // m).reshape(m, ks, d // m)
    idx = native.DeviceIndex(n, m, dtype=np.uint8)
    idx.set_codebook(book)
    idx.fill_synthetic(seed, scale, 0, n)
    idx.set_doc_rows(np.arange(n + 1, dtype=np.int64), np.arange(n, dtype=np.int32))
    idx.set_psg_rows(np.arange(n, dtype=np.int32))

    def decoded(ids):
        elem = (ids.astype(np.int64)[:, None] * m + np.arange(m)[None, :]).astype(np.uint64)
        codes = ((synth.mix64(elem.reshape(-1), seed) >> np.uint64(32)) % np.uint64(ks)).astype(np.int64).reshape(-1, m)
        return np.concatenate([book[j][codes[:, j]] for j in range(m)], axis=1)

    _full_size_properties(idx, n, 6980, 5000, seed, scale, d, decoded, "PQ 50M")
    idx.close()


@pytest.mark.parametrize("m,ks,longest", [(8, 256, 700), (8, 256, 3000), (8, 256, 5000), (8, 64, 8000), (16, 256, 4500), (8, 256, 100), (8, 256, 9000)])
def test_pq_fused_ordering_equals_the_separate_sort(monkeypatch, m, ks, longest):
    """Exact-ADC rows, one row per candidate: the per-query ordering is folded into the scoring kernel (a workgroup per
    query: candidate -> row -> codes -> look-ups -> dense -> interpolated -> order) for queries of 129 .. 8192 candidates.
    Scores, interpolated scores and the order equal the separate launches (FFX_DEBUG_VARIANT=pq_sort_apart) bit for bit, on
    ragged batches (an empty query, one candidate, exact ties, -0.0), ordered by the interpolated or by the dense score."""
    d = m * 96
    rng = np.random.default_rng(m * 1000 + longest)
    n, nq = 40_000, 9
    codewords = synth.rows(70 + m, 1.0, 0, m * ks, d // m).reshape(m, ks, d // m)
    codes = rng.integers(0, ks, size=(n, m)).astype(np.uint8)
    codes[:50] = codes[0]  # rows with identical codes: exact ties
    queries = synth.rows(71 + m, 1.0, 0, nq, d)
    queries[3] = 0.0       # every score of this query is +-0.0
    idx = native.DeviceIndex(n, m, dtype=np.uint8)
    idx.set_codebook(codewords)
    idx.upload(codes, 0)
    idx.set_doc_rows(np.arange(n + 1), np.arange(n))
    idx.set_psg_rows(np.arange(n))
    lens = [longest, 0, 1, longest // 2, 130, longest - 1, 77, longest, 300]
    pair_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pair_ids = np.concatenate([rng.permutation(n)[:k] for k in lens]).astype(np.int32)
    pair_ids[pair_off[0]:pair_off[0] + 40] = np.arange(40)  # the tied rows, in a known first-stage order
    sparse = (rng.random(pair_ids.shape[0]) * 4).astype(np.float32)
    sparse[pair_off[3]:pair_off[4]] = 0.0  # ... and so is every interpolated score of the all-zero query
    plan = native.Plan(idx)
    plan.set_queries(queries)
    got = {}
    for variant in ("", "pq_sort_apart"):
        if variant:
            monkeypatch.setenv("FFX_DEBUG_VARIANT", variant)
        else:
            monkeypatch.delenv("FFX_DEBUG_VARIANT", raising=False)
        for sp in (sparse, None):
            for sort_ff in (False, True):
                plan.set_pairs(pair_off, pair_ids, sp, 1)
                plan.run(0.3, exact_sum=True, sort=True, sort_ff=sort_ff)
                got[(variant, sp is None, sort_ff)] = plan.fetch(ff=True, interp=sp is not None, order=True) + (plan.stats()["kernel_variant"],)
    monkeypatch.delenv("FFX_DEBUG_VARIANT", raising=False)
    folded = 129 <= max(lens) <= 8192
    for key, (ff, interp, order, variant_no) in got.items():
        if key[0] == "":
            assert variant_no == (2500 + m if folded else 2400 + m), (key, variant_no)
            ff2, interp2, order2, v2 = got[("pq_sort_apart",) + key[1:]]
            assert v2 == 2400 + m
            assert np.array_equal(bits(ff), bits(ff2)) and np.array_equal(order, order2), key
            assert interp is None or np.array_equal(bits(interp), bits(interp2))
            keys = ff if (key[1] or key[2]) else interp
            assert np.array_equal(order.astype(np.int64), fo.order(keys, pair_off)), key
    plan.close()
    idx.close()


@pytest.mark.parametrize("d,m,ks", [(768, 8, 256), (1024, 8, 64), (384, 4, 32), (2048, 16, 16), (96, 1, 8), (256, 2, 200)])
def test_pq_exact_adc_when_subspaces_are_numpy_leaves(d, m, ks):
    """Tabulated leaf sums + balanced pairwise combination == np.sum(q * decode(codes)), bit for bit."""
    rng = np.random.default_rng(d + m)
    n, nq, k = 3000, 5, 700
    codewords = synth.rows(50 + d, 1.0, 0, m * ks, d // m).reshape(m, ks, d // m)
    codes = rng.integers(0, ks, size=(n, m)).astype(np.uint8)
    queries = synth.rows(51 + d, 1.0, 0, nq, d)
    idx = native.DeviceIndex(n, m, dtype=np.uint8)
    idx.set_codebook(codewords)
    idx.upload(codes, 0)
    idx.set_doc_rows(np.arange(n + 1), np.arange(n))
    idx.set_psg_rows(np.arange(n))
    pair_off = np.arange(nq + 1) * k
    pair_ids = np.concatenate([rng.choice(n, size=k, replace=False) for _ in range(nq)]).astype(np.int32)
    sparse = rng.random(nq * k).astype(np.float32)
    ff, interp, _, st = native.score(idx, queries, pair_off, pair_ids, 1, sparse, 0.3, exact_sum=True)
    assert st["kernel_variant"] == 2400 + m  # one row per candidate: candidate -> codes -> look-ups -> interpolation in one kernel
    decoded = np.concatenate([codewords[j][codes[:, j]] for j in range(m)], axis=1)
    ref = np.sum(queries[np.repeat(np.arange(nq), k)] * decoded[pair_ids], axis=1)
    assert np.array_equal(bits(ff), bits(ref)) and np.array_equal(bits(interp), bits(fo.interpolate(sparse, ref, 0.3)))
    # the unfused form (row list + look-up kernel + interpolation kernel) gives the same bits
    import os

    os.environ["FFX_DEBUG_VARIANT"] = "pq_no_fuse"
    try:
        ff2, interp2, _, st2 = native.score(idx, queries, pair_off, pair_ids, 1, sparse, 0.3, exact_sum=True)
    finally:
        del os.environ["FFX_DEBUG_VARIANT"]
    assert st2["kernel_variant"] == 2200 + m and np.array_equal(bits(ff2), bits(ff)) and np.array_equal(bits(interp2), bits(interp))
    # ragged queries and documents whose first row is not their ordinal (FIRSTP through doc_first)
    perm = rng.permutation(n).astype(np.int32)
    idx.set_doc_rows(np.arange(n + 1), perm)
    ragged_off = np.array([0, 1, 1, 400, 1300, 2000], dtype=np.int64)
    ids3 = rng.integers(0, n, size=2000).astype(np.int32)
    ff3, _, _, st3 = native.score(idx, queries, ragged_off, ids3, 3, exact_sum=True)
    assert st3["kernel_variant"] == 2400 + m
    ref3 = np.sum(queries[np.repeat(np.arange(nq), np.diff(ragged_off))] * decoded[perm[ids3]], axis=1)
    assert np.array_equal(bits(ff3), bits(ref3))
    idx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("slice_pairs", [0, 1, 70, 400])
def test_rank_kept_equals_score_ranked_against_the_oracle(monkeypatch, slice_pairs):
    """FFX_FLAG_KEEP_DENSE + ffx_rank_kept (Ranking.interpolate on an Index.__call__ result, ranking.py:293-326 after
    index/base.py:389-469): the keeping call returns what it returns without the flag; ffx_rank_kept then gives, for any
    first-stage scores and alpha, the bits of a full ffx_score_ranked -- and of the oracle: interpolated scores in final
    order, the carried words permuted along, the order itself.  Ragged batch with empty queries, every mode over
    multi-passage documents, one slice and many, ordinary and page-locked result buffers; the token is bound to the
    candidate lists and dies with the next keeping call or ffx_index_drop_kept."""
    d, n = 768, 900
    vectors = synth.rows(21, 0.05, 0, n, d)
    queries = synth.rows(22, 0.05, 0, 9, d)
    idx = native.DeviceIndex(n, d)
    idx.upload(vectors, 0)
    rng = np.random.default_rng(5)
    per_doc = rng.integers(1, 4, size=300)
    doc_off = np.concatenate([[0], np.cumsum(per_doc)])
    doc_rows = rng.permutation(n)[: doc_off[-1]]
    idx.set_doc_rows(doc_off, doc_rows)
    idx.set_psg_rows(np.arange(n))
    lens = [120, 0, 1, 77, 300, 0, 0, 250, 33]
    pair_off = np.concatenate([[0], np.cumsum(lens)])
    pair_ids = np.concatenate([rng.choice(300, size=k, replace=False) for k in lens]).astype(np.int32)
    carry = rng.integers(0, 1 << 30, size=len(pair_ids)).astype(np.int32)
    if slice_pairs:
        monkeypatch.setenv("FFX_DEBUG_VARIANT", f"score_slice_pairs={slice_pairs},no_small_path")
    assert native.kept_token(idx) == 0
    with pytest.raises(native.FFXError):
        native.rank_kept(idx, 1, pair_off, None, 0.0)
    last = 0
    for mode in (1, 2, 3, 4):
        ids = pair_ids if mode != 1 else rng.integers(0, n, size=len(pair_ids)).astype(np.int32)
        want_ff, _ = fo.score(vectors, doc_off, doc_rows, np.arange(n), queries, pair_off, ids, mode)
        dense_order = fo.order(want_ff, pair_off)
        plain = native.score_ranked(idx, queries, pair_off, ids, mode, None, 0.0, carry=carry, want_order=True)
        kept = native.score_ranked(idx, queries, pair_off, ids, mode, None, 0.0, carry=carry, want_order=True, keep=True)
        for a, b in zip(plain[:3], kept[:3]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert np.array_equal(bits(kept[0]), bits(want_ff[dense_order])) and np.array_equal(kept[2].astype(np.int64), dense_order)
        token = native.kept_token(idx)
        assert token not in (0, last)
        last = token
        for alpha in (0.0, 0.3, 1.0):
            sparse = rng.random(len(ids), dtype=np.float32) * 20
            want_in = fo.interpolate(sparse, want_ff, alpha)
            want_or = fo.order(want_in, pair_off)
            for pool in (True, False):
                scores, codes, order, _ = native.rank_kept(idx, token, pair_off, sparse, alpha, want_order=True) if pool else \
                    _rank_kept_into(idx, token, pair_off, sparse, alpha)
                assert np.array_equal(bits(scores), bits(want_in[want_or])), (mode, alpha, pool)
                assert np.array_equal(codes, carry[want_or]) and np.array_equal(order.astype(np.int64), want_or)
            full = native.score_ranked(idx, queries, pair_off, ids, mode, sparse, alpha, carry=carry, want_order=True)
            assert np.array_equal(bits(full[0]), bits(want_in[want_or])) and np.array_equal(full[1], carry[want_or])
            assert native.kept_token(idx) == token  # a call that does not keep leaves the kept scores alone
        # no first-stage scores: the dense scores, ordered; other candidate lists are refused
        scores, codes, order, _ = native.rank_kept(idx, token, pair_off, None, 0.0, want_order=True)
        assert np.array_equal(bits(scores), bits(want_ff[dense_order])) and np.array_equal(codes, carry[dense_order])
        other = pair_off.copy()
        other[4] -= 1  # the same totals, another cut
        with pytest.raises(native.FFXError, match="not the candidate lists"):
            native.rank_kept(idx, token, other, None, 0.0)
        with pytest.raises(native.FFXError, match="not the candidate lists"):
            native.rank_kept(idx, token, pair_off[:-1], None, 0.0)
    # kept without carried words: asking for them is refused, the scores come
    ids = pair_ids
    native.score_ranked(idx, queries, pair_off, ids, 2, None, 0.0, keep=True)
    token = native.kept_token(idx)
    with pytest.raises(native.FFXError, match="out_carry"):
        native.rank_kept(idx, token, pair_off, None, 0.0, want_carry=True)
    want_ff, _ = fo.score(vectors, doc_off, doc_rows, np.arange(n), queries, pair_off, ids, 2)
    scores, _, _, _ = native.rank_kept(idx, token, pair_off, None, 0.0, want_carry=False)
    assert np.array_equal(bits(scores), bits(want_ff[fo.order(want_ff, pair_off)]))
    with pytest.raises(native.FFXError, match="token"):
        native.rank_kept(idx, last, pair_off, None, 0.0, want_carry=False)  # the earlier keeping call's token is dead
    # an empty batch, and dropping what is kept
    empty_off = np.zeros(4, dtype=np.int64)
    native.score_ranked(idx, queries[:3], empty_off, np.zeros(0, dtype=np.int32), 2, None, 0.0, keep=True)
    scores, _, _, _ = native.rank_kept(idx, native.kept_token(idx), empty_off, None, 0.0, want_carry=False)
    assert scores.shape == (0,)
    native.drop_kept(idx)
    assert native.kept_token(idx) == 0
    idx.close()


def _rank_kept_into(idx, token, pair_off, sparse, alpha):
    """ffx_rank_kept into page-locked buffers of the caller (the asynchronous result stream)."""
    import ctypes as C

    n = int(pair_off[-1])
    scores, codes, order = native.pinned_empty(max(n, 1), np.float32), native.pinned_empty(max(n, 1), np.int32), native.pinned_empty(max(n, 1), np.int32)
    off = np.ascontiguousarray(pair_off, dtype=np.int64)
    sp = np.ascontiguousarray(sparse, dtype=np.float32)
    st = native.Stats()
    native._check(native.lib().ffx_rank_kept(idx._h, int(token), off.shape[0] - 1, native._ptr(off), native._ptr(sp), float(alpha), native.FLAG_SORT,
                                             native._ptr(scores), native._ptr(codes), native._ptr(order), C.byref(st)))
    return scores[:n], codes[:n], order[:n], st.as_dict()


@pytest.mark.parametrize("slice_pairs", [1, 50, 400])
def test_one_shot_score_is_independent_of_slicing(monkeypatch, slice_pairs):
    """ffx_score pipelines slices of whole queries through two plans: scores, interpolation, the
    per-query order (global positions) and the position of an unknown id must not depend on where
    the batch is cut; page-locked buffers (allocated or locked in place) give the same bytes."""
    d, n = 768, 900
    vectors = synth.rows(21, 0.05, 0, n, d)
    queries = synth.rows(22, 0.05, 0, 9, d)
    idx = native.DeviceIndex(n, d)
    idx.upload(vectors, 0)
    rng = np.random.default_rng(5)
    per_doc = rng.integers(1, 4, size=300)
    doc_off = np.concatenate([[0], np.cumsum(per_doc)])
    doc_rows = rng.permutation(n)[: doc_off[-1]]
    idx.set_doc_rows(doc_off, doc_rows)
    idx.set_psg_rows(np.arange(n))
    lens = [120, 0, 1, 77, 300, 0, 0, 250, 33]
    pair_off = np.concatenate([[0], np.cumsum(lens)])
    pair_ids = np.concatenate([rng.choice(300, size=k, replace=False) for k in lens]).astype(np.int32)
    sparse = rng.random(len(pair_ids), dtype=np.float32) * 20
    for mode in (2, 3, 4):
        want_ff, want_in = fo.score(vectors, doc_off, doc_rows, np.arange(n), queries, pair_off, pair_ids, mode, sparse, 0.3)
        monkeypatch.setenv("FFX_DEBUG_VARIANT", f"score_slice_pairs={slice_pairs}")
        ff, interp, order, st = native.score(idx, queries, pair_off, pair_ids, mode, sparse, 0.3, sort=True)
        assert np.array_equal(bits(ff), bits(want_ff)) and np.array_equal(bits(interp), bits(want_in))
        assert np.array_equal(order.astype(np.int64), fo.order(want_in, pair_off))
        assert st["n_pairs"] == len(pair_ids) and st["n_queries"] == 9
        # page-locked in place, and allocated page-locked
        out = tuple(native.pinned_empty(len(pair_ids), t) for t in (np.float32, np.float32, np.int32))
        with native.pinned(queries, pair_ids, sparse):
            ff2, in2, or2, _ = native.score(idx, queries, pair_off, pair_ids, mode, sparse, 0.3, sort=True, out=out)
        assert np.array_equal(bits(ff2), bits(ff)) and np.array_equal(bits(in2), bits(interp)) and np.array_equal(or2, order)
        assert ff2.base is not None  # views of the page-locked block
    # arrays large enough to be locked in place (native.pinned skips those under 1 MiB)
    monkeypatch.setenv("FFX_DEBUG_VARIANT", "score_slice_pairs=60000")
    big_off = np.arange(10, dtype=np.int64) * 40_000
    big_ids = rng.integers(0, 300, size=int(big_off[-1])).astype(np.int32)
    big_sp = rng.random(len(big_ids), dtype=np.float32)
    want_ff, want_in = fo.score(vectors, doc_off, doc_rows, np.arange(n), queries, big_off, big_ids, 2, big_sp, 0.1)
    with native.pinned(big_ids, big_sp) as pin:
        assert len(pin._done) == 2
        ff, interp, _, _ = native.score(idx, queries, big_off, big_ids, 2, big_sp, 0.1)
    assert np.array_equal(bits(ff), bits(want_ff)) and np.array_equal(bits(interp), bits(want_in))
    monkeypatch.setenv("FFX_DEBUG_VARIANT", f"score_slice_pairs={slice_pairs}")
    bad = pair_ids.copy()
    bad[600] = 10**6
    with pytest.raises(native.NotFound) as e:
        native.score(idx, queries, pair_off, bad, 2)
    assert e.value.pair == 600
    # the index is still usable after a failed call
    ff, _, _, _ = native.score(idx, queries, pair_off, pair_ids, 2)
    want_ff, _ = fo.score(vectors, doc_off, doc_rows, np.arange(n), queries, pair_off, pair_ids, 2)
    assert np.array_equal(bits(ff), bits(want_ff))
    idx.close()


def test_rows_move_into_the_fastest_of_a_few_allocations():
    """ffx_index_reseat (``DeviceIndex.reseat``, ``InMemoryIndex.consolidate``): candidates are allocated, timed with a random gather
    of whole rows, the rows end up in the fastest one and everything scores to the same bits as before; a small index is
    left where it is; what was measured comes back."""
    n, d = 90_000, 768  # 276 MB: above the size below which nothing is tried
    vectors = synth.rows(71, 0.05, 0, n, d)
    idx = native.DeviceIndex(n, d)
    idx.upload(vectors, 0)
    idx.set_doc_rows(np.arange(n + 1), np.arange(n))
    idx.set_psg_rows(np.arange(n))
    rng = np.random.default_rng(3)
    nq, k = 6, 700
    queries = synth.rows(72, 0.05, 0, nq, d)
    pair_off = np.arange(nq + 1, dtype=np.int64) * k
    pair_ids = np.concatenate([rng.choice(n, size=k, replace=False) for _ in range(nq)]).astype(np.int32)
    sparse = rng.random(nq * k, dtype=np.float32)
    plan = native.Plan(idx)  # (a plan made BEFORE the move keeps working: the rows' address is read at launch)
    plan.set_queries(queries)
    plan.set_pairs(pair_off, pair_ids, sparse, 2)
    plan.run(0.2, exact_sum=True, sort=True)
    before = plan.fetch(interp=True, order=True)
    where = idx.device_ptr()
    rep = idx.reseat()  # (the default: as many candidates as half of the free memory holds, at most 7 -- 276 MB a block on an empty card)
    assert rep["candidates"] == 8 and 0 <= rep["chosen"] < 8 and rep["bytes"] == vectors.nbytes
    assert len(rep["probe_ms"]) == 8 and all(ms > 0 for ms in rep["probe_ms"]) and rep["probe_ms"][rep["chosen"]] == min(rep["probe_ms"])
    assert (idx.device_ptr() != where) == (rep["chosen"] != 0)
    assert 4 <= rep["settle_s"] < 10 and rep["probe_ms_settled"] > 0  # (the slow spell after the losers are freed is waited out)
    plan.run(0.2, exact_sum=True, sort=True)
    after = plan.fetch(interp=True, order=True)
    for x, y in zip(before, after):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    want, want_i = fo.score(vectors, np.arange(n + 1), np.arange(n), np.arange(n), queries, pair_off, pair_ids, 2, sparse, 0.2)
    assert np.array_equal(bits(after[0]), bits(want)) and np.array_equal(bits(after[1]), bits(want_i))
    pick = rng.choice(n, size=500, replace=False)
    assert np.array_equal(idx.gather(pick), vectors[pick])
    assert idx.reseat(0)["candidates"] == 1
    assert 0 < idx.probe_placement() < 100  # (the probe alone, as tools/placement_probe.py uses it: milliseconds)
    plan.close()
    idx.close()
    small = native.DeviceIndex(1000, 64)
    assert small.reseat(3) == {"candidates": 1, "chosen": 0, "probe_ms": [0.0], "bytes": 1000 * 64 * 4, "settle_s": 0.0, "probe_ms_settled": 0.0}
    with pytest.raises(native.FFXError):
        small.reseat(8)
    small.close()
    # the API: consolidate() is where the reference says "done adding" (index/memory.py:121-131)
    from fast_forward_amd.encoder import LambdaEncoder
    from fast_forward_amd.index import InMemoryIndex

    api = InMemoryIndex(LambdaEncoder(lambda t: queries[int(t)]), init_size=2000)
    api.add(vectors[:2000], doc_ids=[f"D{i}" for i in range(2000)])
    api.consolidate()  # (a store of this size stays where it is: the wiring is what is looked at -- the wait above is not paid twice)
    assert api.last_placement == {"candidates": 1, "chosen": 0, "probe_ms": [0.0], "bytes": 2000 * d * 4, "settle_s": 0.0, "probe_ms_settled": 0.0}
    api.close()


def test_ordinary_memory_travels_through_the_librarys_own_blocks(monkeypatch):
    """csrc/host_xfer.h: arrays in ordinary memory above 1 MiB -- index rows, candidate lists, first-stage scores, result
    buffers, one slice or many, one staging thread or several -- are cut into pieces that go through page-locked blocks of
    the library (ffx_host_xfer_stats counts them), never to the runtime, which would page-lock the caller's pages on the fly;
    page-locked buffers are handed over as they are (the counters stand still).  Same bytes either way, against the oracle."""
    d, n_docs = 768, 5000
    rng = np.random.default_rng(77)
    vectors = synth.rows(31, 0.05, 0, n_docs, d)  # 15 MB: four pieces, two staging threads
    before = native.xfer_stats()
    idx = native.DeviceIndex(n_docs, d)
    idx.upload(vectors, 0)
    after = native.xfer_stats()
    assert after["transfers"] == before["transfers"] + 1 and after["bytes"] - before["bytes"] == vectors.nbytes
    idx.set_doc_rows(np.arange(n_docs + 1), np.arange(n_docs))
    idx.set_psg_rows(np.arange(n_docs))
    back = idx.gather(np.arange(n_docs)) if hasattr(idx, "gather") else None
    if back is not None:
        assert np.array_equal(back, vectors)
    # sizes around the pieces (1, 1, 2, 2, 4, 4, 8, 8 ... MiB from the graded end): a last piece of a few bytes, an odd number of pieces on one thread, pieces dealt to two threads
    for n_rows in (342, 683, 1366, 1367, 2733, 5500, 11000):
        rows = synth.rows(40 + n_rows, 0.05, 0, n_rows, d)
        odd = native.DeviceIndex(n_rows, d)
        odd.upload(rows, 0)
        assert np.array_equal(odd.gather(np.arange(n_rows)), rows), n_rows
        odd.close()
    n_q, k = 40, 20_000  # 800 k candidates: 3.2 MB per array (nothing here takes the single-submission path)
    queries = synth.rows(32, 0.05, 0, n_q, d)
    pair_off = np.arange(n_q + 1, dtype=np.int64) * k
    pair_ids = rng.integers(0, n_docs, size=n_q * k).astype(np.int32)
    sparse = rng.random(n_q * k, dtype=np.float32) * 10
    want_ff, want_in = fo.score(vectors, np.arange(n_docs + 1), np.arange(n_docs), np.arange(n_docs), queries, pair_off, pair_ids, 2, sparse, 0.2)
    want_order = fo.order(want_in, pair_off)
    for variant in ("no_small_path", "no_small_path,score_slice_pairs=300000", "no_small_path,score_slice_pairs=300000,score_blocking_out"):
        monkeypatch.setenv("FFX_DEBUG_VARIANT", variant)
        at = native.xfer_stats()
        ff, interp, order, _ = native.score(idx, queries, pair_off, pair_ids, 2, sparse, 0.2, sort=True)
        moved = native.xfer_stats()
        assert np.array_equal(bits(ff), bits(want_ff)) and np.array_equal(bits(interp), bits(want_in)), variant
        assert np.array_equal(order.astype(np.int64), want_order), variant
        # candidate ordinals and first-stage scores went up through the blocks, slice by slice (a slice's arrays of at
        # most 1 MiB -- the short slices a call begins and ends with -- go to the runtime, which copies those itself)
        share = 0.9 if variant == "no_small_path" else 0.3
        assert moved["bytes"] - at["bytes"] >= share * (pair_ids.nbytes + sparse.nbytes), (variant, at, moved)
    # the staging blocks can be given back; they return with the next staged transfer
    native.xfer_release()
    ff, interp, order, _ = native.score(idx, queries, pair_off, pair_ids, 2, sparse, 0.2, sort=True)
    assert np.array_equal(bits(ff), bits(want_ff)) and np.array_equal(order.astype(np.int64), want_order)
    # page-locked buffers both ways: nothing is staged
    out = tuple(native.pinned_empty(n_q * k, t) for t in (np.float32, np.float32, np.int32))
    with native.pinned(queries, pair_ids, sparse):
        at = native.xfer_stats()
        ff2, in2, or2, _ = native.score(idx, queries, pair_off, pair_ids, 2, sparse, 0.2, sort=True, out=out)
        assert native.xfer_stats() == at
    assert np.array_equal(bits(ff2), bits(want_ff)) and np.array_equal(bits(in2), bits(want_in)) and np.array_equal(or2.astype(np.int64), want_order)
    idx.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float16, np.uint8])
def test_upload_rows_gathers_from_chunk_maps(monkeypatch, dtype):
    """ffx_index_upload_rows: stored rows scattered over equally sized chunks land in consecutive
    device rows, through staging blocks smaller than the request (several round trips)."""
    rng = np.random.default_rng(8)
    d, cs = 96, 50
    if dtype == np.uint8:
        chunks = [rng.integers(0, 4, size=(cs, d), dtype=np.uint8) for _ in range(4)]  # codes of a Ks = 4 quantizer (others are refused)
    else:
        chunks = [rng.standard_normal((cs, d)).astype(dtype) for _ in range(4)]
    stored = np.concatenate(chunks)
    want = np.concatenate([rng.permutation(4 * cs), rng.integers(0, 4 * cs, size=333)])
    idx = native.DeviceIndex(len(want) + 7, d, dtype=dtype)
    if dtype == np.uint8:
        idx.set_codebook(rng.standard_normal((d, 4, 2)).astype(np.float32))
    monkeypatch.setenv("FFX_DEBUG_VARIANT", f"stage_bytes={40 * d * np.dtype(dtype).itemsize}")  # 40 rows per staging block
    idx.upload_rows(chunks, want, 7, threads=3)
    got = idx.gather(np.arange(7, 7 + len(want)))
    assert np.array_equal(got, stored[want])
    idx.upload_rows(chunks, want[:0], 0)  # nothing to do
    with pytest.raises(native.FFXError):
        idx.upload_rows(chunks, np.array([4 * cs]), 0)  # past the stored rows
    with pytest.raises(native.FFXError):
        idx.upload_rows(chunks, want, 8)  # past the capacity
    with pytest.raises(ValueError):
        idx.upload_rows([c.astype(np.float64) for c in chunks], want, 0)
    idx.close()


LENGTHS = [0, 1, 2, 7, 8, 9, 63, 64, 65, 511, 512, 513, 1000, 0, 4095, 4096, 4097, 5000, 8191, 8192]


def test_per_query_order():
    """The per-query order against the oracle: every power-of-two boundary up to 8192, empty
    queries, heavy ties (ties keep first-stage order), +-0.0 equal, negative scores, infinities,
    and a query longer than 8192."""
    rng = np.random.default_rng(12)
    pair_off = np.concatenate([[0], np.cumsum(LENGTHS)]).astype(np.int64)
    n = int(pair_off[-1])
    scores = rng.integers(-3, 4, size=n).astype(np.float32)          # seven distinct values: ties everywhere
    smooth = rng.random(n) < 0.4
    scores[smooth] = rng.standard_normal(int(smooth.sum())).astype(np.float32) * 5
    zeros = np.flatnonzero(scores == 0)
    scores[zeros[::2]] = -0.0
    scores[rng.integers(0, n, 5)] = np.inf
    scores[rng.integers(0, n, 5)] = -np.inf
    seg_off = np.arange(n + 1, dtype=np.int64)
    ff, _, order = native.aggregate(scores, seg_off, 2, pair_off=pair_off, sort=True)
    assert np.array_equal(bits(ff), bits(scores))
    assert np.array_equal(order.astype(np.int64), fo.order(scores, pair_off))
    # with sparse scores the interpolated score is the key
    sparse = rng.integers(0, 3, size=n).astype(np.float32)
    _, interp, order = native.aggregate(scores, seg_off, 2, sparse, 0.5, pair_off=pair_off, sort=True)
    finite = np.isfinite(interp)
    assert finite.sum() >= n - 10
    assert np.array_equal(order.astype(np.int64), fo.order(interp, pair_off))
    # a very long query (device-wide sort on (query, score) words instead of one workgroup per query)
    huge_off = np.array([0, 5, 5, 5 + 70_001, 5 + 70_001 + 4096, 5 + 70_001 + 4096 + 1], dtype=np.int64)
    huge = rng.integers(-20, 20, size=int(huge_off[-1])).astype(np.float32)
    huge[rng.random(huge.shape[0]) < 0.5] *= np.float32(0.37)
    huge[huge == 0][::2] = -0.0
    huge_sparse = rng.integers(0, 2, size=huge.shape[0]).astype(np.float32)
    _, h_interp, order = native.aggregate(huge, np.arange(huge_off[-1] + 1, dtype=np.int64), 2, huge_sparse, 0.25, pair_off=huge_off, sort=True)
    assert np.array_equal(order.astype(np.int64), fo.order(h_interp, huge_off))
    # a long query between short ones
    long_off = np.array([0, 3, 3 + 8193, 3 + 8193 + 100], dtype=np.int64)
    long_scores = rng.integers(0, 50, size=int(long_off[-1])).astype(np.float32)
    _, _, order = native.aggregate(long_scores, np.arange(long_off[-1] + 1, dtype=np.int64), 2, pair_off=long_off, sort=True)
    assert np.array_equal(order.astype(np.int64), fo.order(long_scores, long_off))


def test_c_example_runs_against_the_oracle(tmp_path):
    """tests/c_abi/rerank.c: the ABI used from plain C99 (no Python in the process) -- index build,
    ffx_score with MaxP + interpolation + order, error reporting -- self-checked against the
    oracle library bit for bit."""
    import shutil
    import subprocess
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    lib_dir, oracle_dir = root / "fast-forward-indexes_amd" / "fast_forward_amd", root / "oracle"
    fo.lib()  # makes sure libffx_oracle.so is built
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler on this box")
    exe = tmp_path / "rerank"
    build = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", f"-I{root / 'include'}",
                            str(root / "tests" / "c_abi" / "rerank.c"), "-o", str(exe), f"-L{lib_dir}", "-lffx", f"-L{oracle_dir}",
                            "-lffx_oracle", "-lm", f"-Wl,-rpath,{lib_dir}", f"-Wl,-rpath,{oracle_dir}", "-Wl,-rpath,/opt/rocm/lib"],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout[-1000:] + run.stderr[-3000:]
    assert "rerank.c ok: 1200 pairs, 3600 rows gathered" in run.stdout


@pytest.mark.parametrize("args,ranks", [(["rccl"], 1), (["host", "2"], 2), (["host", "3"], 3)])
def test_sharded_c_example_runs_against_the_oracle(tmp_path, args, ranks):
    """tests/c_abi/rerank_sharded.c: the multi-GPU entry points (ffx_comm_*, ffx_sharded_*) driven from plain
    C99 -- no Python, no PyTorch in the processes: one rank over RCCL, and 2 / 3 forked ranks sharing this
    GPU over the shared-memory transport (communicator id through a pipe); every owner checks its
    queries against the oracle bit for bit."""
    import shutil
    import subprocess
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    lib_dir, oracle_dir = root / "fast-forward-indexes_amd" / "fast_forward_amd", root / "oracle"
    fo.lib()
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler on this box")
    exe = tmp_path / "rerank_sharded"
    build = subprocess.run(["gcc", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-Wall", "-Wextra", "-pedantic", "-Werror", f"-I{root / 'include'}",
                            str(root / "tests" / "c_abi" / "rerank_sharded.c"), "-o", str(exe), f"-L{lib_dir}", "-lffx", f"-L{oracle_dir}",
                            "-lffx_oracle", "-lm", f"-Wl,-rpath,{lib_dir}", f"-Wl,-rpath,{oracle_dir}", "-Wl,-rpath,/opt/rocm/lib"],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([str(exe), *args], capture_output=True, text=True, timeout=300, env=dict(os.environ, FFX_COMM_TIMEOUT_S="60"))
    assert run.returncode == 0, run.stdout[-1000:] + run.stderr[-3000:]
    assert f"rerank_sharded.c ok: {ranks} rank(s)" in run.stdout and "702 owned pairs checked" in run.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,dim,shift", [(np.float32, 768, 0), (np.float32, 1100, 0), (np.float32, 100, 0), (np.float32, 8, 0), (np.float32, 4, 0),
                                             (np.float32, 3, 0), (np.float32, 768, 4), (np.float32, 768, 8), (np.float32, 1100, 8), (np.float16, 768, 0), (np.float16, 5, 0), (np.float16, 64, 2),
                                             (np.uint8, 8, 0), (np.uint8, 5, 0), (np.uint8, 96, 1), (np.uint8, 700, 3)])
def test_device_pull_of_rows_out_of_page_locked_chunks(dtype, dim, shift):
    """ffx_index_pull_rows_async: the device copies scattered rows out of page-locked host chunks itself.  Every unit
    width the kernel chooses (16-, 8-, 4-byte, single bytes: by row length and by where the chunks start), rows shorter
    than a wave (several rows per wave) and longer than one pass, chunk counts that do not divide the row list, a
    destination offset, an empty list, a row outside the chunks, and chunks that are NOT page-locked (refused)."""
    rng = np.random.default_rng(dim * 7 + shift)
    item = np.dtype(dtype).itemsize
    chunk_rows, n_chunks = 37, 5
    chunk_bytes = chunk_rows * dim * item
    stride = -(-(chunk_bytes + shift) // 64) * 64
    block = native.pinned_empty(n_chunks * stride + 64, np.uint8)
    block[:] = rng.integers(0, 256, size=block.shape[0], dtype=np.uint8)
    chunks = [block[c * stride + shift:c * stride + shift + chunk_bytes].view(dtype).reshape(chunk_rows, dim) for c in range(n_chunks)]
    n_src = chunk_rows * n_chunks
    dev = native.DeviceIndex(400, dim, dtype=dtype)
    try:
        for n, at in ((0, 0), (1, 0), (63, 3), (64, 0), (65, 11), (300, 100)):
            rows = rng.integers(0, n_src, size=n)
            dev.pull_rows(chunks, rows, at)
            dev.upload_wait()
            got = dev.gather(np.arange(at, at + n, dtype=np.int32))
            want = np.stack([chunks[r // chunk_rows][r % chunk_rows] for r in rows]) if n else np.zeros((0, dim), dtype=dtype)
            assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (n, at)
        # the host rewrites the chunks (an index file that is appended to): the next pull sees the new bytes, whatever
        # lines of the old ones an earlier kernel left in the caches
        block[:] = ~block
        rows = rng.integers(0, n_src, size=200)
        dev.pull_rows(chunks, rows, 0)
        dev.upload_wait()
        want = np.stack([chunks[r // chunk_rows][r % chunk_rows] for r in rows])
        assert np.array_equal(dev.gather(np.arange(200, dtype=np.int32)).view(np.uint8), want.view(np.uint8))
        with pytest.raises(native.FFXError, match="outside"):
            dev.pull_rows(chunks, np.array([n_src]), 0)
        with pytest.raises(native.FFXError, match="capacity"):
            dev.pull_rows(chunks, np.zeros(10, dtype=np.int64), 395)
        ordinary = [np.array(c) for c in chunks]
        with pytest.raises(native.FFXError, match="page-locked"):
            dev.pull_rows(ordinary, np.array([0]), 0)
    finally:
        dev.close()


@pytest.mark.gpu
def test_pulls_of_two_windows_stay_ordered_when_a_row_list_grows():
    """ffx_index_upload_after between two windows, before and after one of them needs a longer row list than it has
    page-locked room for (the list's block is replaced; the ordering event must survive that)."""
    rng = np.random.default_rng(5)
    chunk_rows, dim = 50_000, 4
    block = native.pinned_empty((2 * chunk_rows, dim), np.float32)
    block[:] = rng.standard_normal(block.shape, dtype=np.float32)
    chunks = [block[:chunk_rows], block[chunk_rows:]]
    a, b = native.DeviceIndex(90_000, dim), native.DeviceIndex(90_000, dim)
    try:
        for n in (100, 80_000, 300, 90_000):
            rows_a, rows_b = rng.integers(0, 2 * chunk_rows, size=n), rng.integers(0, 2 * chunk_rows, size=n)
            a.pull_rows(chunks, rows_a, 0)
            b.upload_after(a)
            b.pull_rows(chunks, rows_b, 0)
            a.upload_after(b)
            a.upload_wait()
            b.upload_wait()
            assert np.array_equal(a.gather(np.arange(n, dtype=np.int32)), block[rows_a])
            assert np.array_equal(b.gather(np.arange(n, dtype=np.int32)), block[rows_b])
    finally:
        a.close()
        b.close()


@pytest.mark.gpu
def test_randomised_configurations_against_the_oracle():
    """tests/fuzz_score.py (the soak tool: ragged batches, every mode, 27 row lengths, both stores, random slicing of
    ffx_score, sort on and off, every result compared with the oracle bit for bit), 80 configurations of one fixed seed."""
    import subprocess
    import sys
    from pathlib import Path

    run = subprocess.run([sys.executable, str(Path(__file__).resolve().parent / "fuzz_score.py"), "--rounds", "80", "--seed", "11"],
                         capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and "fuzz ok: 80 configurations" in run.stdout, run.stdout[-2000:] + run.stderr[-3000:]
