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
