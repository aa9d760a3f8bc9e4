"""``fast_forward_amd.Ranking`` against the behaviour the reference's own tests pin
(reference tests/test_ranking.py:23-208) and against frames produced by the reference
(tests/golden/interpolate.npz).  CPU only."""

import tempfile
from pathlib import Path

import numpy as np
import pandas as pd
import pytest

from _cases import ALPHAS, GOLDEN, bits
from fast_forward_amd import Ranking

RUN = {"q1": {"d0": 1, "d1": 2, "d2": 300}, "q2": {"d0": 4, "d1": 5, "d2": 600, "d3": 7}}
QUERIES = {"q1": "query 1", "q2": "query 2"}


@pytest.fixture
def plain():
    return Ranking.from_run(RUN)


@pytest.fixture
def with_queries():
    return Ranking.from_run(RUN, queries=QUERIES)


def test_container_protocol(plain):
    assert plain.q_ids == {"q1", "q2"} and len(plain) == 2
    assert "q1" in plain and "q2" in plain and "q3" not in plain
    assert set(iter(plain)) == {"q1", "q2"}
    assert plain["q1"] == {"d2": 300.0, "d1": 2.0, "d0": 1.0}
    assert plain._df["score"].dtype == np.float32
    # q_id descending, then score descending
    assert plain._df["q_id"].tolist() == ["q2"] * 4 + ["q1"] * 3
    assert plain._df["id"].tolist() == ["d2", "d3", "d1", "d0", "d2", "d1", "d0"]


def test_queries_attached_to_the_right_ids(plain, with_queries):
    assert not plain.has_queries and with_queries.has_queries
    df = with_queries._df
    assert set(df.loc[df["q_id"] == "q1", "query"]) == {"query 1"}
    assert set(df.loc[df["q_id"] == "q2", "query"]) == {"query 2"}
    with pytest.raises(ValueError):
        Ranking.from_run(RUN, queries={"qx": "other"})
    more = dict(QUERIES, qx="other")
    assert Ranking.from_run(RUN, queries=more) == with_queries
    assert plain.attach_queries(QUERIES).has_queries
    with pytest.raises(ValueError):
        plain.attach_queries({"q1": "only one"})


def test_equality_ignores_order_and_empty_queries():
    a = Ranking.from_run({"q1": {"d1": 1, "d2": 2}})
    assert a == Ranking.from_run({"q1": {"d2": 2, "d1": 1}})
    assert a != Ranking.from_run({"q1": {"d1": 2, "d2": 3}})
    assert a == Ranking.from_run({"q1": {"d1": 1, "d2": 2}, "q2": {}})
    assert a != "string" and a != 0


def test_operators(plain, with_queries):
    assert plain + 0 == plain and plain * 1 == plain
    assert plain + 1 == Ranking.from_run({"q1": {"d0": 2, "d1": 3, "d2": 301}, "q2": {"d0": 5, "d1": 6, "d2": 601, "d3": 8}})
    assert plain * 2 == Ranking.from_run({"q1": {"d0": 2, "d1": 4, "d2": 600}, "q2": {"d0": 8, "d1": 10, "d2": 1200, "d3": 14}})
    assert 1 + plain == plain + 1 and 2 * plain == plain * 2 and plain * 2 == plain + plain
    for r in (with_queries + 1, with_queries * 2, with_queries + plain, plain + with_queries):
        assert r.has_queries
    # a pair missing on one side counts as 0
    other = Ranking.from_run({"q1": {"d0": 1, "d3": 1}, "q3": {"d0": 1}})
    assert plain + other == Ranking.from_run(
        {"q1": {"d0": 2, "d1": 2, "d2": 300, "d3": 1}, "q2": {"d0": 4, "d1": 5, "d2": 600, "d3": 7}, "q3": {"d0": 1}})
    with pytest.raises(TypeError):
        plain + "x"
    with pytest.raises(TypeError):
        plain * "x"


def test_cut(plain, with_queries):
    assert plain.cut(2) == Ranking.from_run({"q1": {"d2": 300, "d1": 2}, "q2": {"d2": 600, "d3": 7}})
    assert with_queries.cut(2).has_queries
    assert plain.cut(100) == plain


def test_save_load_roundtrip(plain):
    plain.name = "Dummy"
    with tempfile.TemporaryDirectory() as tmp:
        f = Path(tmp) / "sub" / "run.tsv"
        plain.save(f)
        back = Ranking.from_file(f)
        first = f.read_text().splitlines()[0].split("\t")
    assert back == plain and back.name == "Dummy"
    assert first == ["q2", "Q0", "d2", "1", "600.0", "Dummy"]


def test_normalize(with_queries):
    got = Ranking.from_run({"q1": {"d0": 1, "d1": 2, "d2": 3}, "q2": {"d0": 4, "d1": 5, "d2": 6}}).normalize()
    assert got == Ranking.from_run({"q1": {"d0": 0, "d1": 1 / 5, "d2": 2 / 5}, "q2": {"d0": 3 / 5, "d1": 4 / 5, "d2": 1}})
    assert Ranking.from_run({"q1": {"d0": 5, "d1": 5}}).normalize() == Ranking.from_run({"q1": {"d0": 0, "d1": 0}})
    assert with_queries.normalize().has_queries


def test_interpolate_known_answers(plain, with_queries):
    df = with_queries._df.copy()
    df["score"] = np.arange(len(df), dtype=np.float32)
    r_int = plain.interpolate(Ranking(df), 0.5)
    assert r_int != plain and r_int.has_queries
    assert r_int["q1"] == {"d2": 152.0, "d1": 3.5, "d0": 3.5}
    assert r_int["q2"] == {"d2": 300.0, "d3": 4.0, "d1": 3.5, "d0": 3.5}
    r3 = Ranking.from_run({"q1": {"d1": 1, "d2": 2}})
    assert r3.interpolate(r3, 0.5, normalize=True) == Ranking.from_run({"q1": {"d1": 0, "d2": 1}})
    r4 = Ranking.from_run({"q1": {"d1": 1, "d2": 1}, "q2": {"d0": 1}})
    r5 = Ranking.from_run({"q1": {"d0": 1, "d1": 1}, "q3": {"d0": 1}})
    assert r4.interpolate(r5, 0.5) == Ranking.from_run({"q1": {"d0": 0.5, "d1": 1, "d2": 0.5}, "q2": {"d0": 0.5}, "q3": {"d0": 0.5}})
    assert r4.interpolate(r5, 0.5) == 0.5 * r4 + 0.5 * r5


def test_rr_scores(plain, with_queries):
    assert plain.rr_scores(k=1) == Ranking.from_run(
        {"q1": {"d0": 1 / 4, "d1": 1 / 3, "d2": 1 / 2}, "q2": {"d0": 1 / 5, "d1": 1 / 4, "d2": 1 / 2, "d3": 1 / 3}})
    assert with_queries.rr_scores().has_queries


def test_duplicate_pair_rejected():
    df = pd.DataFrame([["q1", "d1", 1.0], ["q1", "d1", 2.0], ["q1", "d2", 1.0]], columns=["q_id", "id", "score"])
    with pytest.raises(ValueError):
        Ranking(df)


def test_nan_scores_dropped_and_ids_become_strings():
    df = pd.DataFrame({"q_id": [1, 1, 2], "id": [10, 11, 12], "score": [1.0, np.nan, 3.0]})
    r = Ranking(df)
    assert r["1"] == {"10": 1.0} and r["2"] == {"12": 3.0}


def _frame(z, prefix):
    return pd.DataFrame({"q_id": z[f"{prefix}_q_id"], "id": z[f"{prefix}_id"], "score": z[f"{prefix}_score"]})


@pytest.mark.parametrize("alpha", ALPHAS)
@pytest.mark.parametrize("norm", [False, True])
def test_interpolate_matches_reference_bits_and_order(alpha, norm):
    """Frames a and b only partly overlap; expected frames were written by the reference."""
    z = np.load(GOLDEN / "interpolate.npz")
    a, b = Ranking(_frame(z, "a")), Ranking(_frame(z, "b"))
    got = a.interpolate(b, alpha, normalize=norm)._df
    key = f"int{alpha}_{'norm' if norm else 'raw'}"
    exp = _frame(z, key)
    want = {(q, i): s for q, i, s in zip(exp["q_id"], exp["id"], exp["score"])}
    have = {(q, i): s for q, i, s in zip(got["q_id"], got["id"], got["score"])}
    assert have.keys() == want.keys()
    ks = sorted(want)
    assert np.array_equal(bits(np.array([have[k] for k in ks])), bits(np.array([want[k] for k in ks])))
    # same ordering: q_id sequence identical, scores non-increasing inside each query
    assert got["q_id"].tolist() == exp["q_id"].tolist()
    assert np.array_equal(bits(got["score"].to_numpy()), bits(exp["score"].to_numpy()))


def _general_interpolate(a: Ranking, b: Ranking, alpha: float, normalize: bool) -> Ranking:
    """The pandas route (outer merge on the string pairs) the integer-key join replaces."""
    from fast_forward_amd.ranking import _scaled_to_unit_range

    left = _scaled_to_unit_range(a._df) if normalize else None
    right = _scaled_to_unit_range(b._df) if normalize else None
    both = a._joined(b, left, right)
    both["score"] = alpha * both["score"] + (1 - alpha) * both["score_other"]
    return a._derived(both, is_sorted=False)


@pytest.mark.parametrize("seed", range(6))
def test_integer_key_join_equals_the_pandas_merge(seed):
    """interpolate / + through `_combined`: same rows in the same order with the same bits as the
    outer merge + constructor, for overlapping and disjoint pairs, exact ties, queries present on
    either side, float64 on one side, and normalisation."""
    rng = np.random.default_rng(seed)
    docs = [f"d{i}" for i in range(40)]

    def random_ranking(q_ids, dtype, coarse):
        run = {}
        for q in q_ids:
            picked = rng.choice(docs, size=int(rng.integers(0, 25)), replace=False)
            # coarse scores make exact ties (within and across the two sides) common
            run[q] = {d: float(rng.integers(0, 4)) if coarse else float(rng.standard_normal()) for d in picked}
        return Ranking.from_run(run, dtype=np.dtype(dtype))

    qs_a, qs_b = [f"q{i}" for i in range(0, 9)], [f"q{i}" for i in range(3, 12)]
    a = random_ranking(qs_a, np.float32, coarse=seed % 2 == 0)
    b = random_ranking(qs_b, np.float64 if seed == 3 else np.float32, coarse=seed % 2 == 0)
    if seed in (1, 4):
        a = a.attach_queries({q: f"text {q}" for q in qs_a})
    if seed in (2, 4):
        b = b.attach_queries({q: f"text {q}" for q in qs_b})
    for alpha, normalize in ((0.1, False), (0.5, False), (0.3, True)):
        fast = a.interpolate(b, alpha, normalize=normalize)
        want = _general_interpolate(a, b, alpha, normalize)
        assert list(fast._df.columns) == list(want._df.columns)
        for col in ("q_id", "id", "query"):
            if col in want._df.columns:
                assert fast._df[col].tolist() == want._df[col].tolist(), col
        assert fast._df["score"].dtype == want._df["score"].dtype
        assert np.array_equal(fast._df["score"].to_numpy().view(np.uint32), want._df["score"].to_numpy().view(np.uint32))
        assert fast.q_ids == want.q_ids and fast == want
        # the cached numbering of the assembled result is what the constructor would compute
        q_no, q_order = pd.factorize(fast._df["q_id"])
        assert np.array_equal(fast._numbering()[0], q_no) and fast._numbering()[1] == list(q_order)
    total = a + b
    both = a._joined(b)
    both["score"] = both["score"] + both["score_other"]
    want = a._derived(both, is_sorted=False)
    assert total._df["id"].tolist() == want._df["id"].tolist() and total == want


def test_save_of_a_large_ranking_is_byte_identical_to_pandas(tmp_path, monkeypatch):
    """Rankings of >= 4096 float32 rows are written by libffx's threaded formatter: the same bytes
    as DataFrame.to_csv (ranks, float formatting, query order), and anything a CSV writer would
    have to quote falls back to pandas."""
    rng = np.random.default_rng(7)
    n_q, k = 37, 400
    scores = rng.standard_normal(n_q * k).astype(np.float32) * rng.choice([1e-6, 1.0, 1e5, 1e17], size=n_q * k).astype(np.float32)
    scores[:5] = [0.0, -0.0, 1.0, 16777216.0, 1e-5]
    df = pd.DataFrame({"q_id": np.repeat([f"q{i:03d}" for i in rng.permutation(n_q)], k),
                       "id": [f"doc-{i}" for i in rng.permutation(n_q * k)], "score": scores})
    r = Ranking(df, name="run-ß")
    fast, slow = tmp_path / "a" / "fast.tsv", tmp_path / "slow.tsv"
    r.save(fast)
    monkeypatch.setattr(Ranking, "_save_native", lambda self, target: False)
    r.save(slow)
    assert fast.read_bytes() == slow.read_bytes()
    assert Ranking.from_file(fast) == Ranking.from_file(slow)
    monkeypatch.undo()
    # ids that need quoting, an unnamed ranking ("None"), float64 scores: pandas writes them, same result either way
    odd = df.copy()
    odd.loc[17, "id"] = "has\ttab"
    odd.loc[99, "id"] = 'has"quote'
    for ranking in (Ranking(odd, name="x"), Ranking(df), Ranking(df, name="f64", dtype=np.dtype(np.float64))):
        a, b = tmp_path / "n.tsv", tmp_path / "p.tsv"
        ranking.save(a)
        monkeypatch.setattr(Ranking, "_save_native", lambda self, target: False)
        ranking.save(b)
        monkeypatch.undo()
        assert a.read_bytes() == b.read_bytes()


def _run_lines(rng, n_q, k, id_fmt="D{}", q_fmt="q{:03d}", name="bm25"):
    lines = []
    for q in rng.permutation(n_q):
        docs = rng.choice(10**6, size=k, replace=False)
        scores = np.sort(rng.random(k).astype(np.float32) * 30)[::-1]
        scores[rng.integers(0, k, 3)] = scores[0]  # exact ties
        lines += [f"{q_fmt.format(q)} Q0 {id_fmt.format(d)} {r + 1} {s} {name}" for r, (d, s) in enumerate(zip(docs, scores))]
    return lines


@pytest.mark.parametrize("case", ["text ids", "integer ids", "leading zeros", "na word", "nan score", "quotes", "short line",
                                  "numeric name", "float ids", "mixed ids", "tabs and blank lines"])
def test_from_file_native_parser_equals_the_pandas_route(tmp_path, monkeypatch, case):
    """Large run files go through libffx's threaded parser when that provably gives what
    read_csv + the constructor give, and through pandas otherwise: either way the Ranking is the
    same frame, row for row, with the same name."""
    rng = np.random.default_rng(len(case))
    kwargs = {"integer ids": dict(id_fmt="{}", q_fmt="{}"), "leading zeros": dict(id_fmt="0{}"), "numeric name": dict(name="7"),
              "float ids": dict(id_fmt="{}.50"), "mixed ids": dict(id_fmt="{}")}.get(case, {})
    lines = _run_lines(rng, 40, 400, **kwargs)
    if case == "na word":
        lines[1234] = lines[1234].replace(lines[1234].split()[2], "NA")
    if case == "nan score":
        parts = lines[777].split()
        parts[4] = "nan"
        lines[777] = " ".join(parts)
    if case == "quotes":
        lines[50] = lines[50].replace(" Q0 ", ' Q0 "') .replace(" 51 ", '" 51 ')
    if case == "short line":
        lines[9000] = " ".join(lines[9000].split()[:5])
    if case == "mixed ids":
        lines[5] = lines[5].replace(lines[5].split()[2], "doc-five")
    text = "\n".join(lines) + "\n"
    if case == "tabs and blank lines":
        text = text.replace(" ", "\t", 3000).replace("\n", "\n\n", 7)
    path = tmp_path / "run.tsv"
    path.write_text(text)
    assert path.stat().st_size > 1 << 18
    calls = []
    real = Ranking._from_file_native.__func__

    def spy(cls, f, queries, dtype):
        out = real(cls, f, queries, dtype)
        calls.append(out is not None)
        return out

    monkeypatch.setattr(Ranking, "_from_file_native", classmethod(spy))
    got = Ranking.from_file(path)
    monkeypatch.setattr(Ranking, "_from_file_native", classmethod(lambda cls, f, queries, dtype: None))
    want = Ranking.from_file(path)
    native_expected = case in ("text ids", "integer ids", "nan score", "mixed ids", "tabs and blank lines")
    assert calls == [native_expected], case
    pd.testing.assert_frame_equal(got._df, want._df)
    assert str(got.name) == str(want.name) and got.q_ids == want.q_ids and got == want
    assert np.array_equal(got._numbering()[0], want._numbering()[0]) and got._numbering()[1] == want._numbering()[1]
    if native_expected:
        queries = {q: f"text of {q}" for q in got.q_ids}
        monkeypatch.setattr(Ranking, "_from_file_native", classmethod(real))
        with_q = Ranking.from_file(path, queries=queries)
        pd.testing.assert_frame_equal(with_q._df, want.attach_queries(queries)._df)
        with pytest.raises(ValueError):
            Ranking.from_file(path, queries={"only": "one"})


@pytest.mark.parametrize("token,raises", [("0x1A", True), ("0x1p3", True), ("nan(123)", True), ("NAN", False), ("1e", True), ("--1", True)])
def test_from_file_native_parser_leaves_odd_score_spellings_to_pandas(tmp_path, monkeypatch, token, raises):
    """strtod would read hexadecimal floats and "nan(...)" as numbers; pandas.read_csv does not.  The
    native parser refuses such files, so from_file behaves exactly as the pandas route does (the
    reference raises ValueError on them, or keeps an upper-case NAN as a NaN-scored row)."""
    rng = np.random.default_rng(11)
    lines = _run_lines(rng, 40, 400)
    parts = lines[4321].split()
    parts[4] = token
    lines[4321] = " ".join(parts)
    path = tmp_path / "run.tsv"
    path.write_text("\n".join(lines) + "\n")
    assert Ranking._from_file_native(path, None, np.dtype(np.float32)) is None
    if raises:
        with pytest.raises(ValueError):
            Ranking.from_file(path)
    else:
        got = Ranking.from_file(path)
        monkeypatch.setattr(Ranking, "_from_file_native", classmethod(lambda cls, f, queries, dtype: None))
        pd.testing.assert_frame_equal(got._df, Ranking.from_file(path)._df)


def test_from_file_native_parser_reports_repeated_pairs(tmp_path):
    rng = np.random.default_rng(1)
    lines = _run_lines(rng, 40, 400)
    lines.append(lines[10])
    path = tmp_path / "dup.tsv"
    path.write_text("\n".join(lines) + "\n")
    with pytest.raises(ValueError, match="Only one score per query-document/passage pair"):
        Ranking.from_file(path)


def test_cut_on_arrays_equals_the_frame_route():
    from fast_forward_amd.ranking import _ranks_within_query

    rng = np.random.default_rng(0)
    run = {f"q{q}": {f"d{d}": float(rng.integers(0, 5)) for d in rng.choice(50, size=int(rng.integers(0, 30)), replace=False)}
           for q in range(12)}
    r = Ranking.from_run(run, queries={f"q{q}": f"t{q}" for q in range(12)})
    for cutoff in (0, 1, 3, 100):
        got = r.cut(cutoff)
        top = r._df[_ranks_within_query(r._df) <= cutoff].reset_index(drop=True)
        want = Ranking(top, name=r.name, dtype=np.dtype(np.float32), copy=True, is_sorted=True)
        pd.testing.assert_frame_equal(got._df, want._df)
        assert got.q_ids == want.q_ids and got._numbering()[1] == want._numbering()[1]
        assert np.array_equal(got._numbering()[0], want._numbering()[0])


@pytest.mark.parametrize("seed", range(8))
def test_column_array_constructor_equals_the_general_one(monkeypatch, seed):
    """Frames of >= 2048 rows with string ids take the constructor's integer-key route: same frame
    (rows, order, dtypes), same numbering, same errors as the general pandas route, for sorted and
    unsorted input, missing scores, attached or separate queries, float64 and int scores."""
    rng = np.random.default_rng(seed)
    n_q, k = 30, 100
    rows = []
    for q in rng.permutation(n_q):
        for d in rng.choice(5000, size=k, replace=False):
            rows.append((f"q{q:02d}", f"doc{d}", float(rng.integers(0, 6)) if seed % 2 else float(rng.standard_normal())))
    df = pd.DataFrame(rows, columns=["q_id", "id", "score"])
    if seed in (2, 3):
        df.loc[rng.choice(len(df), size=50, replace=False), "score"] = np.nan
    if seed == 4:
        df["score"] = rng.integers(0, 100, size=len(df))  # integer scores
    if seed in (5, 6):
        df["query"] = "text of " + df["q_id"]
    if seed == 7:
        df["extra"] = 1.0  # dropped by both routes
    kwargs = dict(name="n", dtype=np.dtype(np.float64) if seed == 6 else np.dtype(np.float32))
    queries = {f"q{q:02d}": f"Q{q}" for q in range(n_q)} if seed in (1, 3) else None
    for is_sorted in (False, True):
        frame = df if not is_sorted else Ranking(df, **kwargs)._df
        fast = Ranking(frame, queries=queries, is_sorted=is_sorted, **kwargs)
        with monkeypatch.context() as m:
            m.setattr(Ranking, "_init_from_columns", lambda self, *a: False)
            slow = Ranking(frame, queries=queries, is_sorted=is_sorted, **kwargs)
        pd.testing.assert_frame_equal(fast._df, slow._df)
        assert fast.q_ids == slow.q_ids and fast._numbering()[1] == slow._numbering()[1]
        assert np.array_equal(fast._numbering()[0], slow._numbering()[0])
    # the input frame is never modified, and copy=True really copies
    before = df.copy()
    r = Ranking(df, copy=True, is_sorted=False, **kwargs)
    pd.testing.assert_frame_equal(df, before)
    dup = pd.concat([df, df.iloc[[17]]])
    with pytest.raises(ValueError, match="Only one score per query-document/passage pair"):
        Ranking(dup, **kwargs)
    assert r._df["score"].dtype == kwargs["dtype"]


@pytest.mark.parametrize("variant", ["shuffled rows", "one object per cell", "shared objects", "two texts for one query", "a cell that is no string",
                                     "a missing id", "float64 scores"])
def test_coded_constructor_equals_the_general_one_on_awkward_frames(monkeypatch, variant):
    """The constructor numbers the text columns natively (object pointers first, then the distinct strings) and keeps
    the result coded.  Whatever the frame looks like -- rows of a query scattered over it, every cell its own string
    object or all cells of a value sharing one, a query id with two texts, cells that are not strings -- the rows,
    their order, the numbering and the errors are those of the general pandas route."""
    rng = np.random.default_rng(11)
    n_q, k = 40, 90
    q_names = np.array([f"q{q:02d}" for q in range(n_q)], dtype=object)
    d_names = np.array([f"doc{d}" for d in range(4000)], dtype=object)
    q_of = np.repeat(rng.permutation(n_q), k)
    d_of = np.concatenate([rng.choice(4000, size=k, replace=False) for _ in range(n_q)])
    score = rng.standard_normal(n_q * k).astype(np.float64 if variant == "float64 scores" else np.float32)
    score[rng.choice(len(score), 30, replace=False)] = np.nan
    cols = {"q_id": q_names[q_of], "id": d_names[d_of], "score": score, "query": np.array(["text " + q for q in q_names], dtype=object)[q_of]}
    if variant == "one object per cell":
        cols = {c: (np.array([(x + " ")[:-1] for x in v.tolist()], dtype=object) if v.dtype == object else v) for c, v in cols.items()}
        assert cols["q_id"][0] is not cols["q_id"][1]
    df = pd.DataFrame(cols)
    if variant == "shuffled rows":
        df = df.iloc[rng.permutation(len(df))].reset_index(drop=True)
    if variant == "two texts for one query":
        df.loc[5, "query"] = "another text"
    if variant == "a cell that is no string":
        df.loc[7, "id"] = 12345
    if variant == "a missing id":
        df.loc[9, "id"] = None
    took = {}
    real = Ranking._init_from_columns
    monkeypatch.setattr(Ranking, "_init_from_columns", lambda self, *a: took.setdefault("fast", real(self, *a)))
    fast = Ranking(df, name="x", dtype=np.dtype(score.dtype))
    assert took["fast"] == (variant not in ("two texts for one query", "a cell that is no string", "a missing id"))
    assert (fast._coded is not None) == took["fast"]
    monkeypatch.setattr(Ranking, "_init_from_columns", lambda self, *a: False)
    slow = Ranking(df, name="x", dtype=np.dtype(score.dtype))
    pd.testing.assert_frame_equal(fast._df, slow._df)
    assert fast._numbering()[1] == slow._numbering()[1] and np.array_equal(fast._numbering()[0], slow._numbering()[0])
    assert fast.has_queries and fast._query_texts() == slow._query_texts()
    assert fast == slow and fast.cut(7) == slow.cut(7)
    monkeypatch.setattr(Ranking, "_init_from_columns", real)
    dup = pd.concat([df, df.iloc[[100]]], ignore_index=True)
    if not np.isnan(dup["score"].iloc[-1]) or True:
        with pytest.raises(ValueError, match="Only one score per query-document/passage pair"):
            Ranking(dup)
    with pytest.raises(ValueError, match="Queries are incomplete"):
        Ranking(df.drop(columns=["query"]), queries={"q00": "only one"})


def test_from_file_coded_equals_the_frame_route(tmp_path, monkeypatch):
    """A large run file is read straight into the coded form (ids numbered natively, every query ordered by host
    threads); rows, order, values, query numbering and everything derived from them -- cut, attach_queries, save,
    ==, the frame itself -- equal what the frame route (small files) gives.  Queries out of order in the file,
    exact ties, +-0 scores, non-ASCII ids, a query with a single row."""
    import fast_forward_amd.ranking as R
    from fast_forward_amd import Ranking

    rng = np.random.default_rng(12)
    lines = []
    q_ids = [f"q{i:03d}" for i in rng.permutation(140)]
    for q in q_ids:
        k = 1 if q == q_ids[5] else int(rng.integers(300, 900))
        docs = rng.choice(5000, size=k, replace=False)
        scores = np.round(rng.standard_normal(k) * 3, 1)  # many exact ties
        scores[rng.random(k) < 0.02] = 0.0
        scores[rng.random(k) < 0.02] = -0.0
        for i, (d, s) in enumerate(zip(docs, scores)):
            lines.append(f"{q}\tQ0\t{'dö' if d % 11 == 0 else 'D'}{d}\t{i + 1}\t{float(s)!r}\tbm25")
    path = tmp_path / "run.tsv"
    path.write_text("\n".join(lines) + "\n", encoding="utf-8")
    assert len(lines) >= R._CODED_FROM_FILE_ROWS and path.stat().st_size >= (1 << 18)
    queries = {q: f"text of {q}" for q in q_ids}

    def both(fn):
        coded = fn()
        monkeypatch.setattr(R, "_CODED_FROM_FILE_ROWS", 10**15)
        frame = fn()
        monkeypatch.undo()
        return coded, frame

    for qs in (None, queries):
        coded, frame = both(lambda: Ranking.from_file(path, queries=qs))
        assert coded._coded is not None and coded._frame is None and frame._coded is None
        assert coded._q_order == frame._q_order and np.array_equal(coded._q_no, frame._q_no) and coded.name == frame.name
        assert coded.has_queries == frame.has_queries == (qs is not None)
        for derive in (lambda r: r, lambda r: r.cut(100), lambda r: r.cut(1), lambda r: r.cut(10**6), lambda r: r.attach_queries(queries), lambda r: r.cut(7).attach_queries(queries)):
            a, b = derive(coded), derive(frame)
            assert a._coded is not None and (a is coded or a._frame is None)  # derived rankings stay coded, no frame is grown
            a.save(tmp_path / "a.tsv")
            b.save(tmp_path / "b.tsv")
            assert a is coded or len(a._q_no) < 4096 or a._frame is None  # (tiny runs are written by pandas)
            assert (tmp_path / "a.tsv").read_bytes() == (tmp_path / "b.tsv").read_bytes()
            assert a._q_order == b._q_order and np.array_equal(a._q_no, b._q_no)
            assert a == b and a._df.equals(b._df)
    with pytest.raises(ValueError, match="Queries are incomplete"):
        Ranking.from_file(path, queries={q: "x" for q in q_ids[:-1]})


def test_from_file_cut_save_match_the_reference_on_a_large_run(tmp_path):
    """Golden: the reference's ``Ranking.from_file(path, queries)``, ``.cut(10)`` and ``.save()`` on a 82 k-row run
    (tests/golden/run_file.npz, oracle/gen_golden.py::run_file_case) against the build's coded route: same rows in the
    same order, same scores (bits), same query texts, and the SAME BYTES written by ``save``."""
    import hashlib

    from _cases import GOLDEN, run_file_text
    from fast_forward_amd import Ranking

    z = np.load(GOLDEN / "run_file.npz")
    text, queries = run_file_text()
    path = tmp_path / "run.tsv"
    path.write_text(text, encoding="utf-8")
    r = Ranking.from_file(path, queries=queries)
    assert r._coded is not None and r._frame is None and str(r.name) == str(z["name"][0])
    for tag, rk in (("full", r), ("cut10", r.cut(10))):
        rk.save(tmp_path / f"{tag}.tsv")
        assert hashlib.sha256((tmp_path / f"{tag}.tsv").read_bytes()).hexdigest() == str(z[f"{tag}_saved_sha256"][0])
        df = rk._df
        assert df["q_id"].tolist() == z[f"{tag}_q_names"][z[f"{tag}_q_code"]].tolist()
        assert df["id"].tolist() == z[f"{tag}_id_names"][z[f"{tag}_id_code"]].tolist()
        assert np.array_equal(df["score"].to_numpy().view(np.uint32), z[f"{tag}_score"].view(np.uint32))
        assert hashlib.sha256("\n".join(df["query"].tolist()).encode()).hexdigest() == str(z[f"{tag}_query"][0])


def test_threaded_outer_join_equals_the_numpy_join_and_the_merge(monkeypatch):
    """Ranking.interpolate / + on two large rankings that are not each other's image: the id tables are merged by
    ffx_strings_code and the outer join runs per query in host threads (ffx_host_outer_join).  Same rows, scores and
    order as the NumPy statement of the join (forced by hiding libffx) and as the reference's own formulation (outer
    merge of the frames, missing -> 0, constructor sort) -- partly overlapping candidate sets, queries that only one
    side has, B's queries in another order, exact ties, query texts on either side."""
    import fast_forward_amd.ranking as R
    from fast_forward_amd import Ranking

    rng = np.random.default_rng(17)
    rows_a, rows_b = [], []
    q_all = [f"q{i:03d}" for i in range(120)]
    for q in q_all:
        pool = rng.choice(30_000, size=700, replace=False)
        in_a, in_b = pool[:500], pool[200:]
        if q != "q007":
            rows_a += [(q, f"D{d}", float(np.round(rng.random() * 20, 1))) for d in in_a]  # one decimal: many exact ties after combining
        if q != "q011":
            rows_b += [(q, f"D{d}", float(np.round(rng.standard_normal(), 1))) for d in in_b]
    rng.shuffle(rows_b)
    fa = pd.DataFrame(rows_a, columns=["q_id", "id", "score"])
    fb = pd.DataFrame(rows_b, columns=["q_id", "id", "score"])
    assert len(fa) + len(fb) >= (1 << 16)
    queries = {q: f"text {q}" for q in q_all}
    for qa, qb in ((queries, None), (None, queries), (None, None)):
        a, b = Ranking(fa, queries=qa, name="a"), Ranking(fb, queries=qb, name="b")
        fast_i, fast_p = a.interpolate(b, 0.3), a + b
        monkeypatch.setattr(R, "_native_or_none", lambda: None)
        a2, b2 = Ranking(fa, queries=qa, name="a"), Ranking(fb, queries=qb, name="b")
        slow_i, slow_p = a2.interpolate(b2, 0.3), a2 + b2
        monkeypatch.undo()
        assert fast_i == slow_i and fast_p == slow_p
        assert fast_i._df.equals(slow_i._df) and fast_p._df.equals(slow_p._df)
        # the reference's formulation
        both = a._df.merge(b._df, on=["q_id", "id"], how="outer", suffixes=(None, "_other")).fillna(0)
        both["score"] = 0.3 * both["score"] + (1 - 0.3) * both["score_other"]
        want = Ranking(both[["q_id", "id", "score"]], name="a")
        assert fast_i == want
        got = fast_i._df
        assert got["q_id"].tolist() == want._df["q_id"].tolist()
        assert np.array_equal(got["score"].to_numpy().view(np.uint32), want._df["score"].to_numpy().view(np.uint32))


def test_large_interpolate_and_add_match_the_reference():
    """Golden (tests/golden/join.npz, oracle/gen_golden.py::join_case): the reference's ``a.interpolate(b, 0.25)`` and
    ``a + b`` for two rankings of 49 k rows each that only partly overlap (a query missing on either side, the second
    frame shuffled, scores that tie exactly after combining) -- through the build's threaded outer join: the same rows,
    the same score bits, and the same order wherever the order is determined (the reference leaves exact ties open;
    here they come out by id, which is also what its sorted merge happens to give)."""
    from _cases import GOLDEN, two_runs
    from fast_forward_amd import Ranking

    z = np.load(GOLDEN / "join.npz")
    fa, fb = two_runs()
    assert len(fa) + len(fb) >= (1 << 16)
    a, b = Ranking(fa, name="a"), Ranking(fb, name="b")
    for tag, got in (("interp", a.interpolate(b, 0.25)), ("plus", a + b)):
        df = got._df
        want_q = z[f"{tag}_q_names"][z[f"{tag}_q_code"]]
        want_id = z[f"{tag}_id_names"][z[f"{tag}_id_code"]]
        want_s = z[f"{tag}_score"]
        assert len(df) == len(want_s)
        assert df["q_id"].tolist() == want_q.tolist()
        assert np.array_equal(df["score"].to_numpy().view(np.uint32), want_s.view(np.uint32))
        have = dict(zip(zip(df["q_id"], df["id"]), df["score"]))
        ref = dict(zip(zip(want_q.tolist(), want_id.tolist()), want_s))
        assert have.keys() == ref.keys() and all(have[k] == ref[k] for k in ref)
        # order: identical outside exact ties; inside a tie both are a permutation of the same ids
        same = np.array(df["id"].tolist()) == want_id
        tied = np.zeros(len(df), dtype=bool)
        eq = (want_s[1:] == want_s[:-1]) & (want_q[1:] == want_q[:-1])
        tied[1:] |= eq
        tied[:-1] |= eq
        assert same[~tied].all()
