"""``fast_forward_amd.quantizer`` against what the reference's tests pin
(reference tests/test_quantizer.py:12-79: equality, dims/dtype, encode/decode shapes,
serialisation round trip, errors) and against the decode-then-score frames the reference produced
with the same codewords (tests/golden/pq_*.npz).  CPU only."""

import numpy as np
import pytest

import ffx_oracle as fo
from _cases import MODE_NO, PQ_CASES, bits, load_pq_case
from fast_forward_amd.quantizer import NanoOPQ, NanoPQ, Quantizer


@pytest.fixture(scope="module", params=[NanoPQ, NanoOPQ])
def pair(request):
    cls = request.param
    rng = np.random.default_rng(0)
    trained = cls(8, 32)
    kw = {"pq_iter": 3, "rotation_iter": 3} if cls is NanoOPQ else {"iter": 3}
    trained.fit(rng.normal(size=(2**9, 768)).astype(np.float32), **kw)
    return cls(8, 32), trained


def test_eq_and_properties(pair):
    fresh, trained = pair
    assert fresh == fresh and trained == trained and fresh != trained and fresh != "x"
    assert fresh.dims == (None, 8) and fresh.dtype == np.uint8 and not fresh._trained
    assert trained.dims == (768, 8) and trained.dtype == np.uint8 and trained._trained


def test_encode_decode_shapes_and_quality(pair):
    _, trained = pair
    x = np.random.default_rng(1).normal(size=(8, 768)).astype(np.float32)
    codes = trained.encode(x)
    assert codes.shape == (8, 8) and codes.dtype == np.uint8
    back = trained.decode(codes)
    assert back.shape == x.shape
    # a quantizer that learned anything reconstructs better than the zero vector
    assert np.linalg.norm(back - x) < np.linalg.norm(x)


def test_serialisation_round_trip(pair):
    fresh, trained = pair
    x = np.random.default_rng(2).normal(size=(8, 768)).astype(np.float32)
    assert Quantizer.deserialize(*fresh.serialize()) == fresh
    loaded = Quantizer.deserialize(*trained.serialize())
    assert loaded == trained and type(loaded) is type(trained)
    assert np.array_equal(trained.encode(x), loaded.encode(x))
    assert np.array_equal(trained.decode(trained.encode(x)), loaded.decode(loaded.encode(x)))


def test_errors(pair):
    fresh, trained = pair
    x = np.zeros((8, 768), dtype=np.float32)
    with pytest.raises(RuntimeError):
        fresh.encode(x)
    with pytest.raises(RuntimeError):
        fresh.decode(np.zeros((8, 8), dtype=np.uint8))
    with pytest.raises(RuntimeError):
        fresh.set_attached()
    clone = Quantizer.deserialize(*trained.serialize())
    clone.set_attached()
    with pytest.raises(RuntimeError):
        clone.fit(x)


def test_opq_rotation_is_orthogonal_and_used():
    rng = np.random.default_rng(3)
    q = NanoOPQ(4, 16)
    x = rng.normal(size=(300, 32)).astype(np.float32)
    q.fit(x, pq_iter=2, rotation_iter=3)
    r = q._R
    assert np.allclose(r @ r.T, np.eye(32), atol=1e-4)
    codes = q.encode(x[:5])
    assert np.allclose(q.decode(codes), q._pq.decode(codes) @ r.T)
    codewords, rot = q.device_codebook()
    assert codewords.shape == (4, 16, 8) and rot is r


def test_code_dtype_widens_with_ks():
    assert NanoPQ(2, 256).dtype == np.uint8 and NanoPQ(2, 257).dtype == np.uint16
    big = NanoPQ(2, 300)
    big.fit(np.random.default_rng(0).normal(size=(400, 8)).astype(np.float32), iter=1)
    assert big.device_codebook() is None  # uint16 codes: decoded rows are stored instead


@pytest.mark.parametrize("name", PQ_CASES)
def test_decode_then_score_matches_reference(name):
    """The oracle on decoded rows reproduces the reference's quantized-index scores bit for bit,
    and NanoPQ.from_codewords encodes/decodes like the quantizer the reference was given."""
    case = load_pq_case(name)
    pq = NanoPQ.from_codewords(case.codewords)
    assert np.array_equal(pq.encode(case.raw_vectors), case.codes)
    assert np.array_equal(bits(pq.decode(case.codes)), bits(case.vectors))
    for mode in case.modes():
        pair_off, pair_ids, sparse, names = case.pairs(mode)
        ff, interp = fo.score(case.vectors, case.doc_off, case.doc_rows, case.psg_row, case.queries, pair_off, pair_ids,
                              MODE_NO[mode], sparse, 0.1)
        pair_q = np.repeat(np.arange(len(case.q_names)), np.diff(pair_off))
        for key, got in ((f"{mode}_ff", ff), (f"{mode}_int0.1", interp)):
            exp = case.expected(key)
            want = np.array([exp[(case.q_names[q], n)] for q, n in zip(pair_q, names)], dtype=np.float32)
            assert np.array_equal(bits(got), bits(want)), (name, key)
