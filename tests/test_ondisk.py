"""``OnDiskIndex``: file behaviour on the CPU (reference tests/test_index.py:513-654: load, stored
quantizer, id length limit, reads across chunks) and streamed scoring on the GPU."""

import shutil

import numpy as np
import pandas as pd
import pytest

from _cases import MODE_NO, assert_same_order_modulo_ties, bits, load_scoring_case
from fast_forward_amd import Ranking
from fast_forward_amd.encoder import LambdaEncoder
from fast_forward_amd.index import InMemoryIndex, Mode, OnDiskIndex
from fast_forward_amd.quantizer import NanoPQ

DOC_IDS = ["d0", "d0", "d1", "d2", "d3"]
PSG_IDS = ["p0", "p1", "p2", "p3", "p4"]
VECTORS = np.tril(np.ones((5, 5), dtype=np.float32))


def _same_vectors(a, b, ids):
    va, ia = a._get_vectors(ids)
    vb, ib = b._get_vectors(ids)
    assert sorted(ia) == sorted(ib)
    assert sorted(map(tuple, np.asarray(va, dtype=np.float64).round(6))) == sorted(map(tuple, np.asarray(vb, dtype=np.float64).round(6)))


def test_create_add_load_roundtrip(tmp_path):
    f = tmp_path / "idx.ffx"
    idx = OnDiskIndex(f, mode=Mode.PASSAGE, chunk_size=2)  # 5 rows span 3 chunk blocks
    assert len(idx) == 0 and idx.dim is None
    idx.add(VECTORS[:3], doc_ids=DOC_IDS[:3], psg_ids=PSG_IDS[:3])
    idx.add(VECTORS[3:], doc_ids=DOC_IDS[3:], psg_ids=PSG_IDS[3:])
    assert len(idx) == 5 and idx.dim == 5 and idx.doc_ids == set(DOC_IDS) and idx.psg_ids == set(PSG_IDS)
    vecs, ids = idx._get_vectors(["p4", "p0", "p2"])
    assert ids == ["p4", "p0", "p2"] and np.array_equal(vecs, VECTORS[[4, 0, 2]])
    idx.mode = Mode.MAXP
    vecs, ids = idx._get_vectors(["d0", "d3"])
    assert ids == ["d0", "d0", "d3"] and np.array_equal(vecs, VECTORS[[0, 1, 4]])
    with pytest.raises(IndexError):
        idx._get_vectors(["nope"])
    with pytest.raises(ValueError):
        OnDiskIndex(f)  # exists, overwrite=False
    shutil.copy(f, tmp_path / "copy.ffx")
    back = OnDiskIndex.load(tmp_path / "copy.ffx", mode=Mode.MAXP)
    assert len(back) == 5 and back.doc_ids == idx.doc_ids and back.psg_ids == idx.psg_ids
    _same_vectors(back, idx, ["d0", "d1", "d2", "d3"])
    back.mode = idx.mode = Mode.PASSAGE
    _same_vectors(back, idx, PSG_IDS)
    for bs in (1, 2, 5, 10):
        v, d, p = zip(*back.batch_iter(bs))
        assert np.array_equal(np.concatenate(v), VECTORS) and sum(d, []) == DOC_IDS and sum(p, []) == PSG_IDS
    # doc-only and psg-only rows survive the round trip as None
    mixed = OnDiskIndex(tmp_path / "mixed.ffx", chunk_size=4)
    mixed.add(VECTORS[:2], doc_ids=["a", "b"])
    mixed.add(VECTORS[2:4], psg_ids=["x", "y"])
    again = OnDiskIndex.load(tmp_path / "mixed.ffx")
    assert again.doc_ids == {"a", "b"} and again.psg_ids == {"x", "y"}
    assert next(again.batch_iter(10))[1:] == (["a", "b", None, None], [None, None, "x", "y"])
    assert len(OnDiskIndex.load(OnDiskIndex(tmp_path / "empty.ffx")._index_file).doc_ids) == 0
    with pytest.raises(ValueError, match="not an index file"):
        (tmp_path / "junk.h5").write_bytes(b"\x89HDF\r\n\x1a\n" + b"0" * 64)
        OnDiskIndex.load(tmp_path / "junk.h5")


def test_max_id_length_leaves_index_unchanged(tmp_path):
    idx = OnDiskIndex(tmp_path / "ids.ffx", max_id_length=3)
    vectors = np.zeros((16, 16))
    doc_ok, psg_ok = ["d1"] * 16, [f"p{i}" for i in range(16)]
    idx.add(vectors, doc_ids=doc_ok, psg_ids=psg_ok)
    with pytest.raises(RuntimeError):
        idx.add(vectors, doc_ids=[d + "-long" for d in doc_ok])
    with pytest.raises(RuntimeError):
        idx.add(vectors, psg_ids=[p + "-long" for p in psg_ok])
    with pytest.raises(RuntimeError):
        idx.add(vectors[:1], psg_ids=["p3"])  # duplicate passage id
    assert idx.doc_ids == set(doc_ok) and idx.psg_ids == set(psg_ok) and len(idx) == 16
    assert len(OnDiskIndex.load(tmp_path / "ids.ffx")) == 16


def test_quantizer_is_stored_and_replaceable_while_empty(tmp_path):
    rng = np.random.default_rng(0)
    first, second = NanoPQ(2, 8), NanoPQ(2, 8)
    first.fit(rng.normal(size=(16, 16)).astype(np.float32))
    second.fit(rng.normal(size=(16, 16)).astype(np.float32))
    idx = OnDiskIndex(tmp_path / "q.ffx")
    idx.quantizer = first
    idx.quantizer = second
    del idx
    back = OnDiskIndex.load(tmp_path / "q.ffx")
    assert back.quantizer == second and back.quantizer != first and back.dim == 16
    x = rng.normal(size=(5, 16)).astype(np.float32)
    back.add(x, doc_ids=DOC_IDS)
    assert back._get_internal_dim() == 2 and len(back) == 5
    codes, _ = back._get_vectors(["d0", "d3"])
    assert codes.dtype == np.uint8 and np.array_equal(codes, second.encode(x)[[0, 1, 4]])
    for vec, _, _ in back:
        assert vec.shape == (16,)
    with pytest.raises(RuntimeError):
        back.quantizer = first  # not empty any more
    again = OnDiskIndex.load(tmp_path / "q.ffx")
    assert again.quantizer == second and np.array_equal(again._get_vectors(["d1"])[0], second.encode(x)[[2]])


def test_reads_across_many_chunks(tmp_path):
    idx = OnDiskIndex(tmp_path / "chunks.ffx", mode=Mode.PASSAGE, init_size=8, chunk_size=4, memory_mapped=True, max_indexing_size=5)
    reps = np.random.default_rng(1).normal(size=(16, 16))
    ids = [f"p{i}" for i in range(16)]
    idx.add(reps, psg_ids=ids)
    vecs, got = idx._get_vectors(ids[::-1])
    assert got == ids[::-1] and np.allclose(vecs, reps[::-1].astype(np.float32))


# ---- GPU: streamed scoring ----------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name,window", [("config1", 40_000), ("d768", 1 << 30), ("d768_f16", 30_000)])
def test_streamed_scores_equal_reference_frames(tmp_path, name, window):
    """Same frames as the reference for an index that never sits in HBM as a whole: with a small
    window the ~2500 candidate rows of config1 go through ~30 windows."""
    case = load_scoring_case(name)
    qvecs = case.queries
    for mode in case.modes():
        idx = OnDiskIndex(tmp_path / f"{name}_{mode}.ffx", LambdaEncoder(lambda t: qvecs[int(t)]), mode=Mode(MODE_NO[mode]),
                          chunk_size=512, window_bytes=window)
        half = len(case.vectors) // 2
        idx.add(case.vectors[:half], doc_ids=case.doc_names[:half], psg_ids=case.psg_names[:half])
        idx.add(case.vectors[half:], doc_ids=case.doc_names[half:], psg_ids=case.psg_names[half:])
        pair_off, _, sparse, names = case.pairs(mode)
        pair_q = np.repeat(np.arange(len(case.q_names)), np.diff(pair_off))
        first = Ranking(pd.DataFrame({"q_id": [case.q_names[q] for q in pair_q], "id": names, "score": sparse}),
                        queries={q: str(i) for i, q in enumerate(case.q_names)})
        dense = idx(first)
        if window < 1 << 20:
            assert idx.last_stats["windows"] > 3
        exp = case.expected_order(f"{mode}_ff")
        for qn in case.q_names:
            rows = dense._df[dense._df["q_id"] == qn]
            assert_same_order_modulo_ties(rows["id"].tolist(), rows["score"].to_numpy(), exp[qn])
        fused = idx.rerank(first, 0.1)
        exp_i = case.expected_order(f"{mode}_int0.1")
        for qn in case.q_names:
            rows = fused._df[fused._df["q_id"] == qn]
            assert_same_order_modulo_ties(rows["id"].tolist(), rows["score"].to_numpy(), exp_i[qn])


@pytest.mark.gpu
def test_to_memory_and_quantized_streaming(tmp_path):
    case = load_scoring_case("d768")
    qvecs = case.queries
    disk = OnDiskIndex(tmp_path / "m.ffx", LambdaEncoder(lambda t: qvecs[int(t)]), mode=Mode.MAXP, chunk_size=100)
    disk.add(case.vectors, doc_ids=case.doc_names, psg_ids=case.psg_names)
    pair_off, _, sparse, names = case.pairs("MAXP")
    pair_q = np.repeat(np.arange(len(case.q_names)), np.diff(pair_off))
    first = Ranking(pd.DataFrame({"q_id": [case.q_names[q] for q in pair_q], "id": names, "score": sparse}),
                    queries={q: str(i) for i, q in enumerate(case.q_names)})
    want = disk(first)
    for mem in (disk.to_memory(), disk.to_memory(batch_size=37)):
        assert isinstance(mem, InMemoryIndex) and mem.doc_ids == disk.doc_ids and mem.psg_ids == disk.psg_ids and len(mem) == len(disk)
        assert mem(first) == want
        mem.close()
    # early stopping and batching run through the streamed scorer too
    assert disk(first, batch_size=3) == want
    # product-quantized file: codes on disk, codes streamed, decode-then-dot on the device
    pq = NanoPQ(8, 16)
    pq.fit(case.vectors, iter=3)
    qdisk = OnDiskIndex(tmp_path / "pq.ffx", LambdaEncoder(lambda t: qvecs[int(t)]), quantizer=pq, mode=Mode.MAXP, chunk_size=64, window_bytes=400)
    qdisk.add(case.vectors, doc_ids=case.doc_names, psg_ids=case.psg_names)
    qmem = InMemoryIndex(LambdaEncoder(lambda t: qvecs[int(t)]), quantizer=NanoPQ.from_codewords(pq._pq.codewords), mode=Mode.MAXP)
    qmem.add(case.vectors, doc_ids=case.doc_names, psg_ids=case.psg_names)
    got, ref = qdisk(first), qmem(first)
    assert got == ref and qdisk.last_stats["windows"] > 3 and qdisk.last_stats["kernel_variant"] == 2108
    loaded = OnDiskIndex.load(tmp_path / "pq.ffx", LambdaEncoder(lambda t: qvecs[int(t)]))
    assert loaded(first) == ref and loaded.to_memory()(first) == ref
