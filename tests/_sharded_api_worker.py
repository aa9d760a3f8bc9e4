"""One rank of the ShardedInMemoryIndex test: N ranks share ONE GPU, scores exchanged through
libffx's shared-memory transport (RCCL needs a GPU per rank); launched by tests/test_index_api.py
as N plain processes (RANK / WORLD_SIZE / FFX_RDZV_KEY in the environment) -- or fewer, several ranks a process, a thread
each (FFX_RANKS_HERE)."""

import sys
from pathlib import Path

import numpy as np
import pandas as pd

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT / "fast-forward-indexes_amd", ROOT / "oracle", ROOT / "tests"):
    sys.path.insert(0, str(p))

from _cases import MODE_NO, assert_same_order_modulo_ties, load_scoring_case  # noqa: E402
from fast_forward_amd import Ranking  # noqa: E402
from fast_forward_amd.encoder import LambdaEncoder  # noqa: E402
from fast_forward_amd.index import Mode  # noqa: E402
from fast_forward_amd.sharded import ShardedInMemoryIndex, connect  # noqa: E402


def main(rank=None, world=None):
    comm = connect("host", rank=rank, world=world)
    rank, world = comm.rank, comm.world
    case = load_scoring_case("config1")
    qvecs = case.queries
    checked = 0
    for mode in ("MAXP", "PASSAGE", "AVEP"):
        idx = ShardedInMemoryIndex(LambdaEncoder(lambda t: qvecs[int(t)]), mode=Mode(MODE_NO[mode]), comm=comm)
        third = len(case.vectors) // 3
        for lo, hi in ((0, third), (third, len(case.vectors))):
            idx.add(case.vectors[lo:hi], doc_ids=case.doc_names[lo:hi], psg_ids=case.psg_names[lo:hi])
        assert len(idx) == len(case.vectors) and 0 < len(idx.local) < len(case.vectors)
        assert idx.doc_ids == {d for d in case.doc_names if d is not None} and idx.psg_ids == {p for p in case.psg_names if p is not None}
        assert idx.dim == case.vectors.shape[1] and idx.quantizer is None and np.array_equal(idx.encode_queries(["1", "0"]), qvecs[[1, 0]])
        pair_off, _, sparse, names = case.pairs(mode)
        pair_q = np.repeat(np.arange(len(case.q_names)), np.diff(pair_off))
        first = Ranking(pd.DataFrame({"q_id": [case.q_names[q] for q in pair_q], "id": names, "score": sparse}),
                        queries={q: str(i) for i, q in enumerate(case.q_names)})
        idx.consolidate()  # (a shard of this size stays where it is; the call is every rank's own)
        dense = idx(first, gather=True)
        exp = case.expected_order(f"{mode}_ff")
        assert dense._df["q_id"].tolist() == case.z[f"{mode}_ff_q_id"].tolist()
        for qn in case.q_names:
            rows = dense._df[dense._df["q_id"] == qn]
            assert_same_order_modulo_ties(rows["id"].tolist(), rows["score"].to_numpy(), exp[qn])
        own = idx.rerank(first, 0.1)  # only the queries this rank owns
        exp_i = case.expected_order(f"{mode}_int0.1")
        assert 0 < len(own.q_ids) < len(case.q_names)
        # what this rank's GPU took in is ITS block of the batch (offsets, id codes + holders + first-stage scores, the owned
        # queries' vectors), not the batch: rows are hash-placed, so the candidates travelled as the ranking's id codes
        st = idx._sharded().stats()
        n_own_q, n_own_p = st["own_queries"], st["own_pairs"]
        assert n_own_q == len(own.q_ids) and 0 < n_own_p < len(names)
        assert st["stage_h2d_bytes"] == (n_own_q + 1) * 8 + n_own_p * 12 + n_own_q * idx.dim * 4, st
        for qn in own.q_ids:
            rows = own._df[own._df["q_id"] == qn]
            assert_same_order_modulo_ties(rows["id"].tolist(), rows["score"].to_numpy(), exp_i[qn])
            checked += len(rows)
        try:
            idx(Ranking.from_run({"q000": {names[0]: 1.0, "missing-id": 2.0}}, queries={"q000": "0"}))
        except IndexError as e:
            assert "missing-id" in str(e)
        else:
            raise AssertionError("unknown id accepted")
        # early stopping over the shards (base.py:316-387): the frames of the one-GPU index, for every rank's owned
        # queries and gathered; an id nobody holds is an error only if the loop reaches it
        from fast_forward_amd.index import InMemoryIndex as _One

        one = _One(LambdaEncoder(lambda t: qvecs[int(t)]), mode=Mode(MODE_NO[mode]))
        one.add(case.vectors, doc_ids=case.doc_names, psg_ids=case.psg_names)
        for cutoff, es_alpha, depths in ((3, 0.2, (5, 20, 100)), (1, 0.5, (1, 4, 30)), (10, 0.0, (10, 50))):
            want = one(first, early_stopping=cutoff, early_stopping_alpha=es_alpha, early_stopping_depths=depths)
            assert idx(first, early_stopping=cutoff, early_stopping_alpha=es_alpha, early_stopping_depths=depths, gather=True) == want
            mine = idx(first, early_stopping=cutoff, early_stopping_alpha=es_alpha, early_stopping_depths=depths)
            assert 0 < len(mine.q_ids) < len(want.q_ids) and all(mine[q] == want[q] for q in mine.q_ids)
            # every depth interval was an owner-routed step: the rank took in candidates of the queries it owns and what it holds of
            # the others' (the last step's figures: at most its share of the batch, never the batch)
            it = idx.last_intake
            assert it["own_pairs"] <= n_own_p and it["held_pairs"] < len(names) and it["batch_pairs"] <= len(names), it
            assert it["bytes"] == (n_own_q + 1) * 8 + it["own_pairs"] * 4 + (len(case.q_names) + it["held_pairs"]) * 4, it
        tail = Ranking.from_run({"q000": {names[0]: 9.0, names[1]: 8.0, "missing-id": 0.5}}, queries={"q000": "0"})
        assert idx(tail, early_stopping=1, early_stopping_alpha=1.0, early_stopping_depths=(2, 3), gather=True) == \
            one(tail, early_stopping=1, early_stopping_alpha=1.0, early_stopping_depths=(2, 3))
        try:
            idx(tail, early_stopping=3, early_stopping_alpha=0.2, early_stopping_depths=(3,), gather=True)
        except IndexError as e:
            assert "missing-id" in str(e)
        else:
            raise AssertionError("unknown id reached and accepted")
        try:
            idx(first, early_stopping=3)
        except ValueError:
            pass
        else:
            raise AssertionError("early stopping without alpha / depths accepted")
        one.close()
        idx.close()
    # a collection whose hash placement leaves the LAST rank without a single row: that rank takes part with empty lists
    # (decided collectively -- a rank raising alone between two collectives would leave the others hanging in the exchange)
    from fast_forward_amd.index import InMemoryIndex  # noqa: E402
    from fast_forward_amd.sharded import _owner_of  # noqa: E402

    keep = [i for i in range(len(case.vectors)) if _owner_of(case.doc_names[i], case.psg_names[i], world) != world - 1][:40]
    sub = ShardedInMemoryIndex(LambdaEncoder(lambda t: qvecs[int(t)]), mode=Mode.MAXP, comm=comm)
    sub.add(case.vectors[keep], doc_ids=[case.doc_names[i] for i in keep], psg_ids=[case.psg_names[i] for i in keep])
    assert (len(sub.local) == 0) == (rank == world - 1)
    whole = InMemoryIndex(LambdaEncoder(lambda t: qvecs[int(t)]), mode=Mode.MAXP)
    whole.add(case.vectors[keep], doc_ids=[case.doc_names[i] for i in keep], psg_ids=[case.psg_names[i] for i in keep])
    docs = sorted({case.doc_names[i] for i in keep})
    run = {f"q{q}": {d: float(1 + (7 * q + j) % 5) for j, d in enumerate(docs) if (q + j) % 2 == 0} for q in range(5)}
    small = Ranking.from_run(run, queries={f"q{q}": str(q) for q in range(5)})
    assert sub(small, gather=True) == whole(small)
    assert sub.rerank(small, 0.3, gather=True) == whole.rerank(small, 0.3)
    assert sub(small, batch_size=7, gather=True) == whole(small)  # the reference's keywords are accepted
    assert sub(small, early_stopping=2, early_stopping_alpha=0.2, early_stopping_depths=(2, 5), gather=True) == \
        whole(small, early_stopping=2, early_stopping_alpha=0.2, early_stopping_depths=(2, 5))  # (one rank holds no rows)
    sub.close()
    whole.close()
    empty = ShardedInMemoryIndex(LambdaEncoder(lambda t: qvecs[int(t)]), mode=Mode.MAXP, comm=comm)
    try:
        empty(small)
    except (RuntimeError, IndexError):
        pass  # on EVERY rank (nobody is left waiting at a collective)
    else:
        raise AssertionError("an index that is empty on every rank scored something")
    empty.close()
    total = int(comm.allreduce([checked])[0])
    assert "torch" not in sys.modules
    if rank == 0:
        print(f"sharded api world={world}: {total} owned rows checked against the reference frames")
    comm.close()


if __name__ == "__main__":
    import _rank_threads

    _rank_threads.run(main)
