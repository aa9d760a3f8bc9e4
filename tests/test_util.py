"""The ``Index.add``-side helpers (fast_forward_amd.util) -- the reference's tests/test_util.py and
tests/test_indexer.py restated, plus create_coalesced_index against vectors produced by the
reference (tests/golden/coalesce.npz, oracle/gen_golden.py::coalesce_case).

The CPU tests drive the helpers through a small host-only stand-in for an index (a test double:
the product indexes keep their rows in HBM); the ``gpu`` tests use the real InMemoryIndex."""

import json
from pathlib import Path

import numpy as np
import pytest

from fast_forward_amd import Ranking
from fast_forward_amd.encoder import LambdaEncoder
from fast_forward_amd.quantizer import NanoPQ
from fast_forward_amd.util import Indexer, cos_dist, create_coalesced_index, to_ir_measures

GOLDEN = Path(__file__).resolve().parent / "golden"


class ListIndex:
    """Host-only test double with the part of the Index interface the helpers touch."""

    def __init__(self):
        self.vectors, self.docs, self.psgs, self._quantizer = [], [], [], None

    def __len__(self):
        return len(self.vectors)

    @property
    def quantizer(self):
        return self._quantizer

    @quantizer.setter
    def quantizer(self, q):
        if len(self) > 0:
            raise RuntimeError("Quantizers can only be attached to empty indexes.")
        self._quantizer = q

    def add(self, vectors, doc_ids=None, psg_ids=None):
        n = len(vectors)
        self.vectors.extend(np.asarray(vectors))
        self.docs.extend(doc_ids if doc_ids is not None else [None] * n)
        self.psgs.extend(psg_ids if psg_ids is not None else [None] * n)

    @property
    def doc_ids(self):
        return {d for d in self.docs if d is not None}

    @property
    def psg_ids(self):
        return {p for p in self.psgs if p is not None}

    def _get_vectors(self, ids):
        rows = [i for want in ids for i, d in enumerate(self.docs) if d == want]
        return np.stack([self.vectors[i] for i in rows]), [self.docs[i] for i in rows]

    def _batch_iter(self, batch_size):
        for lo in range(0, len(self), batch_size):
            yield np.stack(self.vectors[lo:lo + batch_size]), self.docs[lo:lo + batch_size], self.psgs[lo:lo + batch_size]

    batch_iter = _batch_iter


DICTS = [
    {"text": "123", "doc_id": "d1", "psg_id": "d1_p1"},
    {"text": "234", "doc_id": "d1", "psg_id": "d1_p2"},
    {"text": "456", "doc_id": "d1", "psg_id": "d1_p3"},
    {"text": "567", "doc_id": "d2", "psg_id": "d2_p1"},
    {"text": "678", "doc_id": "d3", "psg_id": "d3_p1"},
    {"text": "890", "doc_id": "d4"},
    {"text": "901", "psg_id": "d5_p1"},
]


def _check_from_dicts(make_index):
    target = make_index()
    calls = []

    def encode(text):
        calls.append(text)
        return np.zeros(16, dtype=np.float32)

    Indexer(target, LambdaEncoder(encode), encoder_batch_size=2, batch_size=4).from_dicts(DICTS)
    assert len(target) == 7 and calls == [d["text"] for d in DICTS]
    assert target.doc_ids == {"d1", "d2", "d3", "d4"}
    assert target.psg_ids == {"d1_p1", "d1_p2", "d1_p3", "d2_p1", "d3_p1", "d5_p1"}
    with pytest.raises(RuntimeError):
        Indexer(target, encoder=None).from_dicts(DICTS)


def _check_from_index(make_index):
    source, target = make_index(), make_index()
    source.add(np.arange(16 * 16, dtype=np.float32).reshape(16, 16), doc_ids=[f"d{i}" for i in range(16)])
    Indexer(target, batch_size=5).from_index(source)
    assert source.doc_ids == target.doc_ids and len(target) == 16
    got = np.concatenate([v for v, _, _ in target._batch_iter(16)])
    assert np.array_equal(got, np.arange(16 * 16, dtype=np.float32).reshape(16, 16))


def _check_quantizer(make_index):
    rng = np.random.default_rng(0)
    for fit_batches in (1, 2):
        target = make_index()
        indexer = Indexer(target, encoder=LambdaEncoder(lambda t: rng.normal(size=(32,)).astype(np.float32)),
                          quantizer=NanoPQ(4, 8), batch_size=16, quantizer_fit_batches=fit_batches)
        records = [{"text": f"text_{i}", "doc_id": f"d{i}"} for i in range(64)]
        held = 16 * (fit_batches - 1)  # whole batches that arrive before the quantizer can be fit
        indexer.from_dicts(records[:held])
        assert len(target) == 0 and target.quantizer is None
        indexer.from_dicts(records[held:])
        assert target.quantizer._trained and len(target) == 64
    fitted = NanoPQ(4, 8)
    fitted.fit(rng.normal(size=(64, 64)).astype(np.float32))
    with pytest.raises(ValueError):
        Indexer(make_index(), quantizer=fitted)
    busy = make_index()
    busy.add(np.zeros((8, 16), dtype=np.float32), doc_ids=[f"d{i}" for i in range(8)])
    with pytest.raises(ValueError):
        Indexer(busy, quantizer=NanoPQ(4, 8))


def _check_coalesce(make_index, get_rows):
    case = json.loads((GOLDEN / "manifest.json").read_text())["cases"]["coalesce"]
    z = np.load(GOLDEN / case["file"])
    vectors, doc_ids = z["vectors"], [str(d) for d in z["doc_ids"]]
    for delta in case["deltas"]:
        source, target = make_index(), make_index()
        source.add(vectors, doc_ids=doc_ids)
        create_coalesced_index(source, target, delta, batch_size=50)
        want_v, want_d = z[f"coalesced_{delta}"], [str(d) for d in z[f"coalesced_docs_{delta}"]]
        got_v, got_d = get_rows(target, list(dict.fromkeys(doc_ids)))
        assert got_d == want_d
        assert np.array_equal(np.asarray(got_v, dtype=np.float32).view(np.uint32), want_v.view(np.uint32))
        with pytest.raises(ValueError):
            create_coalesced_index(source, target, delta)  # the target is not empty any more


def test_indexer_from_dicts():
    _check_from_dicts(ListIndex)


def test_indexer_from_index():
    _check_from_index(ListIndex)


def test_indexer_fits_and_attaches_a_quantizer():
    _check_quantizer(ListIndex)


def test_coalesced_index_equals_the_reference():
    _check_coalesce(ListIndex, lambda index, docs: index._get_vectors(docs))


def test_cos_dist_and_ir_measures_frame():
    assert cos_dist(np.array([1.0, 0.0]), np.array([0.0, 2.0])) == 1.0
    assert abs(cos_dist(np.array([1.0, 1.0]), np.array([2.0, 2.0]))) < 1e-12
    with pytest.raises(AssertionError):
        cos_dist(np.zeros((2, 2)), np.zeros(2))
    run = {"q1": {"d0": 1.0, "d1": 3.0}, "q2": {"d2": 2.0}}
    r = Ranking.from_run(run, queries={"q1": "a", "q2": "b"})
    df = to_ir_measures(r)
    assert set(df.columns) == {"query_id", "doc_id", "score"}
    assert df["query_id"].equals(r._df["q_id"]) and df["doc_id"].equals(r._df["id"]) and df["score"].equals(r._df["score"])


# ---- the same through the HBM-resident index ------------------------------------------------------
def _hbm():
    from fast_forward_amd.index import InMemoryIndex

    return InMemoryIndex()


@pytest.mark.gpu
def test_indexer_on_the_hbm_index():
    _check_from_dicts(_hbm)
    _check_from_index(_hbm)
    _check_quantizer(_hbm)


@pytest.mark.gpu
def test_coalesced_hbm_index_equals_the_reference():
    _check_coalesce(_hbm, lambda index, docs: index._get_vectors(docs))


def test_array_encoder_is_a_table_and_encodes_all_queries_in_one_gather():
    """``ArrayEncoder`` (this build's cheapest way to feed pre-computed query vectors): a batch is one gather of the table's
    rows, ``Index.encode_queries`` hands it all texts at once, and the result is what batches of ``encoder_batch_size``
    through an ordinary encoder give (base.py:58-72); the table may grow between calls; an unknown text is a ``KeyError``."""
    import pytest

    from fast_forward_amd.encoder import ArrayEncoder, LambdaEncoder
    from fast_forward_amd.index import InMemoryIndex

    rng = np.random.default_rng(3)
    table = {f"text {i}": rng.standard_normal(24).astype(np.float32) for i in range(100)}
    enc = ArrayEncoder(table)
    asked = [f"text {i}" for i in (5, 99, 0, 5, 42)]
    assert np.array_equal(enc(asked), np.stack([table[t] for t in asked]))
    assert enc([]).shape[0] == 0
    index = InMemoryIndex(enc, encoder_batch_size=7)
    every = [f"text {i}" for i in rng.permutation(100)]
    batched = InMemoryIndex(LambdaEncoder(lambda t: table[t]), encoder_batch_size=7).encode_queries(every)
    got = index.encode_queries(every)
    assert got.dtype == np.float32 and np.array_equal(got, batched)
    table["one more"] = np.ones(24, dtype=np.float32)
    assert np.array_equal(index.encode_queries(["one more", "text 3"]), np.stack([table["one more"], table["text 3"]]))
    with pytest.raises(KeyError):
        enc(["never seen"])
