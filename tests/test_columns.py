"""``Ranking.from_columns`` / ``Ranking.columns`` / ``Names``: the columnar way into and out of a ranking (no Python object
per row) gives what the reference's constructor gives on the same rows (``src/fast_forward/ranking.py:67-121``: repeated pairs
refused, rows without a score dropped, order by query id descending and score descending, queries numbered in order of first
appearance), whatever form the columns come in.  CPU only."""

import numpy as np
import pandas as pd
import pytest

from fast_forward_amd import Ranking
from fast_forward_amd.names import Names

pa = pytest.importorskip("pyarrow")


def _rows(seed: int, n_q: int = 40, k: int = 120, grouped: bool = True):
    rng = np.random.default_rng(seed)
    q_col, id_col, s_col = [], [], []
    for q in rng.permutation(n_q):
        docs = rng.choice(9000, size=k, replace=False)
        q_col += [f"q{q:03d}"] * k
        id_col += [f"doc{d}" if d % 3 else f"ü{d}" for d in docs]  # (a few non-ASCII ids)
        s_col += (rng.integers(0, 6, size=k).astype(np.float32) if seed % 2 else rng.standard_normal(k).astype(np.float32)).tolist()
    q_col, id_col, s_col = np.array(q_col, dtype=object), np.array(id_col, dtype=object), np.array(s_col, dtype=np.float32)
    if not grouped:
        p = rng.permutation(len(q_col))
        q_col, id_col, s_col = q_col[p], id_col[p], s_col[p]
    return q_col, id_col, s_col


def _general(q_col, id_col, s_col, monkeypatch, **kw):
    """The same rows through the constructor's general pandas route (what the reference runs)."""
    with monkeypatch.context() as m:
        m.setattr(Ranking, "_init_from_columns", lambda self, *a: False)
        return Ranking(pd.DataFrame({"q_id": q_col, "id": id_col, "score": s_col}), **kw)


def _forms(q_col, id_col):
    """Every form a text column may come in."""
    q_cat, id_cat = pd.Categorical(q_col), pd.Categorical(id_col)
    ascii_only = all(s.isascii() for s in id_col.tolist())
    forms = {
        "objects": (q_col, id_col),
        "lists": (q_col.tolist(), id_col.tolist()),
        "arrow": (pa.array(q_col, type=pa.string()), pa.array(id_col, type=pa.string())),
        "large arrow": (pa.array(q_col, type=pa.large_string()), pa.array(id_col, type=pa.large_string())),
        "chunked arrow": (pa.chunked_array([pa.array(q_col[:1000]), pa.array(q_col[1000:])]), pa.chunked_array([pa.array(id_col[:77]), pa.array(id_col[77:])])),
        "sliced arrow": (pa.array(np.concatenate([["x"], q_col]), type=pa.string()).slice(1), pa.array(np.concatenate([["y", "z"], id_col]), type=pa.string()).slice(2)),
        "arrow-backed series": (pd.Series(q_col, dtype="string[pyarrow]"), pd.Series(id_col, dtype=pd.ArrowDtype(pa.string()))),
        "categorical": (q_cat, pd.Series(id_cat)),
        "codes + names": ((q_cat.codes, q_cat.categories.to_numpy()), (id_cat.codes.astype(np.int64), pa.array(id_cat.categories.to_numpy(dtype=object)))),
        "bytes": (np.char.encode(q_col.astype("U"), "utf-8"), np.char.encode(id_col.astype("U"), "utf-8")),
    }
    if ascii_only:
        forms["unicode"] = (q_col.astype("U"), id_col.astype("U"))
    return forms


@pytest.mark.parametrize("seed", range(4))
def test_from_columns_equals_the_constructor_for_every_column_form(monkeypatch, seed):
    q_col, id_col, s_col = _rows(seed, grouped=seed != 3)
    queries = {f"q{q:03d}": f"text {q}" for q in range(40)} if seed in (1, 3) else None
    want = _general(q_col, id_col, s_col, monkeypatch, name="n", queries=queries)
    for label, (q_form, id_form) in _forms(q_col, id_col).items():
        got = Ranking.from_columns(q_form, id_form, s_col, name="n", queries=queries)
        assert got._coded is not None, label
        pd.testing.assert_frame_equal(got._df, want._df, obj=label)
        assert got.q_ids == want.q_ids and got._numbering()[1] == want._numbering()[1], label
        assert np.array_equal(got._numbering()[0], want._numbering()[0]), label
        assert got.has_queries == want.has_queries and got.name == "n"


def test_from_columns_checks_what_the_constructor_checks(monkeypatch):
    q_col, id_col, s_col = _rows(5)
    # rows without a score are dropped (ranking.py:103); float64 and integer scores are cast
    holes = s_col.astype(np.float64)
    holes[np.random.default_rng(0).choice(len(holes), size=200, replace=False)] = np.nan
    want = _general(q_col, id_col, holes, monkeypatch)
    got = Ranking.from_columns(pa.array(q_col), pa.array(id_col), holes)
    pd.testing.assert_frame_equal(got._df, want._df)
    ints = np.arange(len(s_col)) % 7
    pd.testing.assert_frame_equal(Ranking.from_columns(q_col, id_col, ints)._df, _general(q_col, id_col, ints, monkeypatch)._df)
    f64 = Ranking.from_columns(q_col, id_col, s_col, dtype=np.dtype(np.float64))
    pd.testing.assert_frame_equal(f64._df, _general(q_col, id_col, s_col, monkeypatch, dtype=np.dtype(np.float64))._df)
    # a repeated (query, id) pair (ranking.py:95-98), incomplete queries (ranking.py:16-28)
    with pytest.raises(ValueError, match="Only one score per query-document/passage pair"):
        Ranking.from_columns(np.append(q_col, q_col[17]), np.append(id_col, id_col[17]), np.append(s_col, 1.0))
    with pytest.raises(ValueError, match="Queries are incomplete"):
        Ranking.from_columns(pa.array(q_col), pa.array(id_col), s_col, queries={"q000": "only this one"})
    # is_sorted: the rows are taken as they stand
    ordered = _general(q_col, id_col, s_col, monkeypatch)._df
    kept = Ranking.from_columns(ordered["q_id"].to_numpy(), ordered["id"].to_numpy(), ordered["score"].to_numpy(), is_sorted=True)
    pd.testing.assert_frame_equal(kept._df, ordered)
    # what is not a column of strings
    for bad in (pa.array([None] + q_col.tolist()[1:]), np.arange(len(q_col)), (np.full(len(q_col), -1), ["a"])):
        with pytest.raises(TypeError):
            Ranking.from_columns(bad, id_col, s_col)
    with pytest.raises(ValueError, match="one entry per row"):
        Ranking.from_columns(q_col[:-1], id_col, s_col)
    with pytest.raises(TypeError):
        Ranking.from_columns(q_col, id_col, np.array(["x"] * len(q_col)))
    empty = Ranking.from_columns([], [], np.zeros(0, dtype=np.float32))
    assert len(empty) == 0 and len(empty._df) == 0


def test_frame_with_arrow_or_categorical_columns_takes_the_columnar_route(monkeypatch):
    """``Ranking(frame)`` on a frame whose text columns are Arrow-backed or categorical: numbered from the buffers, same
    result as the frame of Python strings."""
    q_col, id_col, s_col = _rows(2)
    want = _general(q_col, id_col, s_col, monkeypatch, name="x")
    for frame in (pd.DataFrame({"q_id": pd.Series(q_col, dtype="string[pyarrow]"), "id": pd.Series(id_col, dtype="string[pyarrow]"), "score": s_col}),
                  pd.DataFrame({"q_id": pd.Categorical(q_col), "id": pd.Categorical(id_col), "score": s_col})):
        got = Ranking(frame, name="x")
        assert got._coded is not None
        pd.testing.assert_frame_equal(got._df, want._df)


def test_columns_hands_the_ranking_out_without_a_frame():
    q_col, id_col, s_col = _rows(1)
    r = Ranking.from_columns(pa.array(q_col), pa.array(id_col), s_col)
    cols = r.columns()
    assert r._frame is None  # nothing grew a frame
    assert isinstance(cols["id_names"], Names) and cols["id_codes"].dtype == np.int32 and cols["score"].dtype == np.float32
    frame = r._df
    assert np.array_equal(np.asarray(cols["id_names"])[cols["id_codes"]], frame["id"].to_numpy())
    assert np.array_equal(np.array(cols["q_names"], dtype=object)[cols["q_codes"]], frame["q_id"].to_numpy())
    assert np.array_equal(cols["score"], frame["score"].to_numpy())
    assert np.array_equal(np.diff(cols["pair_off"]), np.bincount(cols["q_codes"]))
    # a ranking that lives on a frame hands out the same columns
    small = Ranking(frame.iloc[:500].copy(), is_sorted=True)
    cs = small.columns()
    assert np.array_equal(np.asarray(cs["id_names"], dtype=object)[cs["id_codes"]], small._df["id"].to_numpy())


def test_names_is_the_array_it_stands_for():
    strings = ["d1", "", "ünï", "d1000", "x y"]
    names = Names("\n".join(strings).encode(), len(strings))
    assert len(names) == 5 and names.shape == (5,) and names.packed == ("\n".join(strings).encode(), 5)
    assert [names[i] for i in range(5)] == strings and names[-1] == "x y"  # single names without making them all
    assert names._objects is None
    with pytest.raises(IndexError):
        names[5]
    assert names[np.array([3, 0])].tolist() == ["d1000", "d1"] and np.asarray(names).tolist() == strings and list(names) == strings
    assert np.concatenate([np.asarray(names), np.asarray(names)]).shape == (10,)
    assert not names.plain_for_csv() and Names(b"a\nb", 2).plain_for_csv() and not Names(b"a\tb\nc", 2).plain_for_csv()
    assert Names(b"", 0).packed is None and len(Names(b"", 1)) == 1 and Names(b"", 1)[0] == ""
    assert Names.joined(Names(b"a\nb", 2), Names(b"c", 1)).tolist() == ["a", "b", "c"]


def test_column_coders_number_every_form_alike():
    """The natives behind the forms: one numbering (it depends on the set of strings only), whatever the layout."""
    from fast_forward_amd import native

    rng = np.random.default_rng(9)
    pool = np.array([f"D{v}" for v in range(3000)] + ["", "ü", "a-rather-long-identifier-of-a-document-000001", "a-rather-long-identifier-of-a-document-000002"], dtype=object)
    col = pool[rng.integers(0, len(pool), size=150_000)]
    ref_codes, blob, n = native.code_strings("\n".join(col.tolist()).encode(), len(col))
    for threads in (1, 5):
        for got in (native.code_arrow_strings(pa.array(col, type=pa.string()), threads), native.code_arrow_strings(pa.array(col, type=pa.large_string()), threads),
                    native.code_fixed_strings(np.char.encode(col.astype("U"), "utf-8"), threads)):
            codes, names = got
            assert np.array_equal(codes, ref_codes) and names.packed == (blob, n)
    # long runs (a query-id column) take the run-head route: same numbers
    runs = np.repeat(pool[rng.permutation(len(pool))[:700]], 300)
    a, names_a = native.code_arrow_strings(pa.array(runs))
    b, blob_b, n_b = native.code_strings("\n".join(runs.tolist()).encode(), len(runs), 1)
    assert np.array_equal(a, b) and names_a.packed == (blob_b, n_b) and np.array_equal(np.asarray(names_a)[a], runs)
    assert native.code_arrow_strings(pa.array(["a", None])) is None and native.code_arrow_strings(pa.array([1, 2])) is None
    assert native.code_arrow_strings(pa.array(["two\nlines", "x"])) is None
    # repeated codes inside a group
    off = np.array([0, 3, 3, 7], dtype=np.int64)
    assert native.repeated_in_groups(np.array([0, 1, 2, 0, 1, 2, 3], dtype=np.int32), off, 4) == -1
    assert native.repeated_in_groups(np.array([0, 1, 2, 0, 1, 2, 1], dtype=np.int32), off, 4) == 2
    with pytest.raises(native.FFXError):
        native.repeated_in_groups(np.array([0, 9], dtype=np.int32), np.array([0, 2], dtype=np.int64), 4)
