"""The CPU oracle against the golden fixtures produced by running the reference
(oracle/gen_golden.py) and against the reference tests' own known answers.  CPU only."""

import json

import numpy as np
import pytest

import ffx_oracle as fo
from _cases import (ALPHAS, F64_ALPHAS, F64_CASES, GOLDEN, MODE_NO, SCORING_CASES, assert_same_order_modulo_ties,
                    assert_within_reference_tolerance, bits, load_scoring_case, load_sub200k, manifest)


@pytest.mark.parametrize("name", SCORING_CASES)
def test_scores_bit_exact(name):
    case = load_scoring_case(name)
    for mode in case.modes():
        pair_off, pair_ids, sparse, names = case.pairs(mode)
        ff, _ = fo.score(case.vectors, case.doc_off, case.doc_rows, case.psg_row, case.queries,
                         pair_off, pair_ids, MODE_NO[mode])
        exp = case.expected(f"{mode}_ff")
        pair_q = np.repeat(np.arange(len(case.q_names)), np.diff(pair_off))
        want = np.array([exp[(case.q_names[q], n)] for q, n in zip(pair_q, names)], dtype=np.float32)
        assert np.array_equal(bits(ff), bits(want)), f"{name}/{mode}: max diff {np.abs(ff - want).max()}"


@pytest.mark.parametrize("name", SCORING_CASES)
def test_numpy_port_matches_c_oracle(name):
    case = load_scoring_case(name)
    for mode in case.modes():
        pair_off, pair_ids, sparse, _ = case.pairs(mode)
        a, ai = fo.score(case.vectors, case.doc_off, case.doc_rows, case.psg_row, case.queries,
                         pair_off, pair_ids, MODE_NO[mode], sparse, 0.1)
        b, bi = fo.numpy_port_score(case.vectors, case.doc_off, case.doc_rows, case.psg_row, case.queries,
                                    pair_off, pair_ids, MODE_NO[mode], sparse, 0.1, batch_queries=3)
        assert np.array_equal(bits(a), bits(b))
        assert np.array_equal(bits(ai), bits(bi))


@pytest.mark.parametrize("name", SCORING_CASES)
def test_pandas_port_matches_c_oracle(name):
    """The pandas-faithful restatement (frames of string ids, merges, group-by, outer merge; what bench.py's cpu_baseline times
    beside the NumPy port) gives the bits of the C oracle -- and so of the reference-run fixtures -- in every mode, whole run
    and in batches of queries (index/base.py:445-459)."""
    case = load_scoring_case(name)
    for mode in case.modes():
        pair_off, pair_ids, sparse, _ = case.pairs(mode)
        if any(len(set(pair_ids[a:b].tolist())) != b - a for a, b in zip(pair_off[:-1], pair_off[1:])):
            continue  # (a frame cannot hold a candidate twice under one query)
        a, ai = fo.score(case.vectors, case.doc_off, case.doc_rows, case.psg_row, case.queries, pair_off, pair_ids, MODE_NO[mode], sparse, 0.1)
        for batch in (None, 3):
            b, bi, steps = fo.pandas_port_score(case.vectors, case.doc_off, case.doc_rows, case.psg_row, case.queries, pair_off, pair_ids, MODE_NO[mode],
                                                sparse, 0.1, batch_queries=batch)
            assert np.array_equal(bits(a), bits(b)) and np.array_equal(bits(ai), bits(bi)), (name, mode, batch)
            assert steps["total_s"] > 0


@pytest.mark.parametrize("name", ["config1", "d768", "s2_sub"])
def test_interpolation_and_order(name):
    case = load_scoring_case(name)
    for mode in case.modes():
        pair_off, pair_ids, sparse, names = case.pairs(mode)
        pair_q = np.repeat(np.arange(len(case.q_names)), np.diff(pair_off))
        for alpha in ALPHAS:
            ff, interp = fo.score(case.vectors, case.doc_off, case.doc_rows, case.psg_row, case.queries,
                                  pair_off, pair_ids, MODE_NO[mode], sparse, alpha)
            exp = case.expected(f"{mode}_int{alpha}")
            want = np.array([exp[(case.q_names[q], n)] for q, n in zip(pair_q, names)], dtype=np.float32)
            assert np.array_equal(bits(interp), bits(want)), f"{name}/{mode}/alpha={alpha}"
            order = fo.order(interp, pair_off)
            exp_order = case.expected_order(f"{mode}_int{alpha}")
            for q, qn in enumerate(case.q_names):
                sel = order[pair_off[q]:pair_off[q + 1]]
                assert_same_order_modulo_ties([names[i] for i in sel], interp[sel], exp_order[qn])


def test_reference_known_answers():
    """tests/test_index.py:135-200 of the reference, through the oracle."""
    ka = manifest()["cases"]["known_answers"]
    vectors = np.array(ka["vectors"], dtype=np.float32)
    doc_names, psg_names = ka["doc_ids"], ka["psg_ids"]
    docs = sorted(set(doc_names), key=doc_names.index)
    doc_rows = [r for d in docs for r, n in enumerate(doc_names) if n == d]
    doc_off = np.cumsum([0] + [doc_names.count(d) for d in docs])
    queries = np.ones((2, 5), dtype=np.float32)
    for mode, exp in ka["expected"].items():
        run = ka["psg_run"] if mode == "PASSAGE" else ka["doc_run"]
        names = psg_names if mode == "PASSAGE" else docs
        pair_ids, pair_off, keys = [], [0], []
        for q in ("q1", "q2"):
            for cid in run[q]:
                pair_ids.append(names.index(cid))
                keys.append((q, cid))
            pair_off.append(len(pair_ids))
        ff, _ = fo.score(vectors, doc_off, doc_rows, np.arange(5), queries, pair_off, pair_ids, MODE_NO[mode])
        for (q, cid), s in zip(keys, ff):
            assert s == exp[q][cid]
    assert ka["expected"]["MAXP"]["q1"]["d0"] == 2 and ka["expected"]["AVEP"]["q1"]["d0"] == 1.5


def test_reference_interpolate_known_answers():
    """tests/test_ranking.py:157-188: alpha=0.5 on integer scores, and missing => 0."""
    s = np.array([300, 2, 1], dtype=np.float32)
    t = np.array([4, 5, 6], dtype=np.float32)
    assert fo.interpolate(s, t, 0.5).tolist() == [152.0, 3.5, 3.5]
    assert fo.interpolate([1, 1, 0], [1, 0, 1], 0.5).tolist() == [1.0, 0.5, 0.5]


def test_interpolate_fixture_pairs_present_on_both_sides():
    z = np.load(GOLDEN / "interpolate.npz")
    a = {(q, i): s for q, i, s in zip(z["a_q_id"].tolist(), z["a_id"].tolist(), z["a_score"])}
    b = {(q, i): s for q, i, s in zip(z["b_q_id"].tolist(), z["b_id"].tolist(), z["b_score"])}
    for alpha in ALPHAS:
        keys = list(zip(z[f"int{alpha}_raw_q_id"].tolist(), z[f"int{alpha}_raw_id"].tolist()))
        assert set(keys) == set(a) | set(b)
        sa = np.array([a.get(k, 0.0) for k in keys], dtype=np.float32)
        sb = np.array([b.get(k, 0.0) for k in keys], dtype=np.float32)
        assert np.array_equal(bits(fo.interpolate(sa, sb, alpha)), bits(z[f"int{alpha}_raw_score"]))


def test_not_found_reports_pair():
    case = load_scoring_case("dim20")
    pair_off, pair_ids, _, _ = case.pairs("MAXP")
    bad = pair_ids.copy()
    bad[7] = len(case.doc_off) + 5
    with pytest.raises(fo.OracleNotFound) as e:
        fo.score(case.vectors, case.doc_off, case.doc_rows, case.psg_row, case.queries, pair_off, bad, 2)
    assert e.value.pair == 7


def test_synth_c_matches_numpy():
    from fast_forward_amd import synth

    for seed, scale, first, n, dim in [(1, 1.0, 0, 7, 5), (9, 0.0625, 12345, 4, 768), (2**40 + 3, 0.3, 2**33, 3, 128)]:
        assert np.array_equal(bits(fo.synth_rows(seed, scale, first, n, dim)), bits(synth.rows(seed, scale, first, n, dim)))


@pytest.mark.parametrize("name", F64_CASES)
def test_fp32_oracle_within_tolerance_of_the_references_float64_path(name):
    """A default reference index larger than one chunk scores in float64 (index/memory.py:106) and
    casts at the end.  The build always follows the single-chunk fp32 semantics; against the float64
    frames the bar is the north star's 1e-4 and an ordering that differs only inside tiny gaps."""
    case = load_scoring_case(name)
    assert int(case.z["chunk"]) == 64 and case.vectors.shape[0] > 3 * 64  # >= 3 reference chunks
    worst = 0.0
    for mode in case.modes():
        pair_off, pair_ids, sparse, names = case.pairs(mode)
        pair_q = np.repeat(np.arange(len(case.q_names)), np.diff(pair_off))
        ff, _ = fo.score(case.vectors, case.doc_off, case.doc_rows, case.psg_row, case.queries, pair_off, pair_ids, MODE_NO[mode])
        exp = case.expected(f"{mode}_ff")
        want = np.array([exp[(case.q_names[q], n)] for q, n in zip(pair_q, names)], dtype=np.float32)
        rep = assert_within_reference_tolerance(ff, want, pair_off, f"{name}/{mode}/dense")
        worst = max(worst, rep["max_abs_diff"] / rep["bar"])
        for alpha in F64_ALPHAS:
            _, interp = fo.score(case.vectors, case.doc_off, case.doc_rows, case.psg_row, case.queries, pair_off, pair_ids,
                                 MODE_NO[mode], sparse, alpha)
            exp_i = case.expected(f"{mode}_int{alpha}")
            want_i = np.array([exp_i[(case.q_names[q], n)] for q, n in zip(pair_q, names)], dtype=np.float32)
            assert_within_reference_tolerance(interp, want_i, pair_off, f"{name}/{mode}/alpha={alpha}")
    assert worst < 0.5, f"fp32 summation uses {worst:.2f} of the tolerance: closer to the bar than expected"


def test_float64_frames_really_differ_from_fp32_frames():
    """multichunk_f64 is the d768 corpus pushed through >= 3 reference chunks: same inputs, different
    arithmetic -- if the two frames were equal the fixture would not exercise the float64 path."""
    a, b = load_scoring_case("d768"), load_scoring_case("multichunk_f64")
    assert np.array_equal(a.vectors, b.vectors) and np.array_equal(a.z["cand_docs"], b.z["cand_docs"])
    ea, eb = a.expected("MAXP_ff"), b.expected("MAXP_ff")
    assert ea.keys() == eb.keys()
    diff = np.array([abs(float(ea[k]) - float(eb[k])) for k in ea])
    assert 0 < diff.max() < 1e-4 * max(1.0, max(abs(float(v)) for v in ea.values()))


def test_sub_sampled_config2_200k_rows_against_both_reference_runs():
    """BASELINE.md section 4's equivalence check: 200 k rows x 768, Q = 200, k = 1000, MaxP.  The oracle
    (and its NumPy port) equal the reference's single-fp32-chunk run bit for bit and sit within the
    1e-4 bar of its default float64-chunk run, with the same ordering outside tiny gaps."""
    from fast_forward_amd import synth
    from _cases import sha, tolerance_report

    c = load_sub200k()
    vectors = synth.rows(c["seed"], c["scale"], 0, c["n_rows"], c["dim"])
    assert sha(vectors) == c["vectors_sha"]
    rows = np.arange(c["n_rows"], dtype=np.int32)
    ff, interp = fo.score(vectors, c["doc_off"], rows, rows, c["queries"], c["pair_off"], c["pair_ids"], MODE_NO["MAXP"], c["sparse"], 0.1)
    assert np.array_equal(bits(ff), bits(c["ff_f32chunk"]))
    rep = assert_within_reference_tolerance(ff, c["ff_default_chunks"], c["pair_off"], "sub200k dense vs float64 chunks")
    assert rep["pairs"] == 200_000 and rep["max_abs_diff"] < 1e-6
    # the NumPy port (what bench.py times as the CPU baseline) on the first queries
    n_q = 8
    hi = int(c["pair_off"][n_q])
    ff_np, _ = fo.numpy_port_score(vectors, c["doc_off"], rows, rows, c["queries"][:n_q], c["pair_off"][:n_q + 1], c["pair_ids"][:hi],
                                   MODE_NO["MAXP"], c["sparse"][:hi], 0.1, batch_queries=4)
    assert np.array_equal(bits(ff_np), bits(c["ff_f32chunk"][:hi]))


@pytest.mark.parametrize("part", ["small", "large", "docs"])
@pytest.mark.parametrize("promotion", ["numpy1", "numpy2"])
def test_early_stopping_statement_matches_the_reference_under_both_numpy_promotions(promotion, part):
    """oracle.early_stopping_mask (base.py:316-387 restated per query) against what the REFERENCE ITSELF kept, run under
    numpy 1.26.4 (its pinned range: the criterion's bound is float64) and under numpy 2.2.6 (float32): same rows, same
    score bits, for the ordinary configurations, for repeated depths (the loop ends), and for the boundary
    configurations (cut-off 1 after a first depth of 1) where the two regimes keep DIFFERENT rows."""
    from _cases import load_es_promotion_part

    c = load_es_promotion_part(promotion, part)
    n_q, lens = len(c["q_names"]), c["lens"]
    pair_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pair_q = np.repeat(np.arange(n_q), lens)
    if part == "docs":
        doc_off = np.concatenate([[0], np.cumsum(np.bincount(c["doc_of_row"]))]).astype(np.int64)
    else:
        doc_off = np.arange(c["n_rows"] + 1, dtype=np.int64)
    rows = np.arange(c["n_rows"], dtype=np.int32)
    differs = 0
    for (mode, cutoff, alpha, depths), want in zip(c["configs"], c["expected"]):
        ff, _ = fo.score(c["vectors"], doc_off, rows, rows, c["queries"], pair_off, c["cands"], MODE_NO[mode])
        kept = fo.early_stopping_mask(ff, c["sparse"], pair_off, cutoff, alpha, depths, promotion)
        have = {(c["q_names"][pair_q[p]], f"{c['prefix']}{c['cands'][p]}"): ff[p] for p in np.flatnonzero(kept)}
        assert have.keys() == want.keys(), (promotion, part, mode, cutoff, alpha, depths)
        ks = sorted(want)
        assert np.array_equal(bits(np.array([have[k] for k in ks])), bits(np.array([want[k] for k in ks], dtype=np.float32)))
        other = fo.early_stopping_mask(ff, c["sparse"], pair_off, cutoff, alpha, depths, "numpy2" if promotion == "numpy1" else "numpy1")
        differs += int((other != kept).any())
    assert differs >= 3  # the fixture does separate the two regimes


def test_fixtures_do_not_depend_on_the_numpy_version_except_early_stopping():
    """The fixtures were generated by the reference under numpy 2.2.6; the reference pins numpy < 2.  The same generator
    run under numpy 1.26.4 (oracle/check_golden_numpy1.py, this image's Python 3.9) reproduced every array of every
    regenerated file bit for bit, except the early-stopping configuration whose continue-criterion flips with the scalar
    promotion -- the case early_stopping_np1.npz / _np2.npz exist for.  This test reads the committed report."""
    import json

    report = json.loads((GOLDEN / "numpy1_check.json").read_text())
    assert report["numpy"].startswith("1.") and len(report["identical"]) >= 24
    assert all(name.startswith("early_stopping") for name in report["differ"])
    for name in ("config1.npz", "d768.npz", "s2_sub.npz", "sub200k.npz", "interpolate.npz", "partial_ids.npz", "d768_f16.npz"):
        assert name in report["identical"], name
