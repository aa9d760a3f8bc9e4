"""One rank of the owner-routed (fresh-batch) sharded path on a GPU shared by all ranks (shared-memory transport), launched by
tests/test_gpu_parity.py as N plain processes (or fewer, with several ranks a process: FFX_RANKS_HERE).  Every rank holds a document range of a synthetic index, is handed ONLY the
block of queries it owns, and checks what comes back against the single-GPU scorer over the whole index (bit for bit):
ffx_sharded_score (sliced, pipelined), stage / run / fetch_async over several different batches, the per-rank intake bytes,
and that an ordinal without rows is the same error on every rank."""

import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT / "fast-forward-indexes_amd", ROOT / "oracle", ROOT / "tests"):
    sys.path.insert(0, str(p))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from _cases import bits  # noqa: E402
from fast_forward_amd import native, synth  # noqa: E402
from fast_forward_amd.sharded import ShardedScorer, connect, doc_partition, query_partition  # noqa: E402


def batch(rng, with_rows, lens):
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ids = np.concatenate([rng.choice(with_rows, size=k, replace=False) for k in lens] + [np.zeros(0, dtype=np.int64)]).astype(np.int32)
    return off, ids, (rng.random(len(ids)) * 9).astype(np.float32)


def main(rank=None, world=None):
    comm = connect("host", rank=rank, world=world, device=0)
    rank, world = comm.rank, comm.world
    rng = np.random.default_rng(77)  # the same stream on every rank: everybody knows the whole batch, and uses its block only
    d, n_docs = 96, 900
    per_doc = rng.integers(0, 4, size=n_docs)  # some documents hold no rows at all
    per_doc[:5] = 1
    doc_off = np.concatenate([[0], np.cumsum(per_doc)]).astype(np.int64)
    n = int(doc_off[-1])
    vectors = synth.rows(31, 0.1, 0, n, d)
    n_q = 11
    queries = synth.rows(32, 0.1, 0, n_q, d)
    full = native.DeviceIndex(n, d)
    full.upload(vectors, 0)
    full.set_doc_rows(doc_off, np.arange(n, dtype=np.int32))
    full.set_psg_rows(np.arange(n, dtype=np.int32))
    bounds = doc_partition(doc_off, world)
    d_lo, d_hi = int(bounds[rank]), int(bounds[rank + 1])
    r_lo, r_hi = int(doc_off[d_lo]), int(doc_off[d_hi])
    shard = native.DeviceIndex(max(1, r_hi - r_lo), d)
    if r_hi > r_lo:
        shard.upload(vectors[r_lo:r_hi], 0)
    shard.set_doc_rows(doc_off[d_lo:d_hi + 1] - r_lo, np.arange(r_hi - r_lo, dtype=np.int32))
    shard.set_psg_rows(np.arange(r_hi - r_lo, dtype=np.int32))
    scorer = ShardedScorer(shard, bounds, comm)
    with_rows = np.flatnonzero(per_doc > 0)
    qb = query_partition(n_q, world)
    q_lo, q_hi = int(qb[rank]), int(qb[rank + 1])
    batches = [batch(rng, with_rows, lens) for lens in ([0, 1, 300, 0, 77, 256, 5, 90, 41, 3, 128], [17] * n_q, [200, 0, 0, 0, 9, 1, 1, 64, 330, 2, 50])]
    checked = 0

    def block(b):
        off, ids, sparse = batches[b]
        lo, hi = int(off[q_lo]), int(off[q_hi])
        return off[q_lo:q_hi + 1] - lo, ids[lo:hi], sparse[lo:hi], lo, hi

    def expected(b, mode, alpha):
        off, ids, sparse = batches[b]
        return native.score(full, queries, off, ids, mode, sparse, alpha, sort=True)

    for mode in (2, 3, 4):
        for b in range(len(batches)):
            own_off, own_ids, own_sp, lo, hi = block(b)
            want_ff, want_in, want_or, _ = expected(b, mode, 0.3)
            for n_slices in (1, 2, 5):
                ff, interp, order = scorer.score_owned(qb, queries[q_lo:q_hi], own_off, own_ids, own_sp, mode, 0.3, sort=True, n_slices=n_slices)
                assert np.array_equal(bits(ff), bits(want_ff[lo:hi])), (rank, mode, b, n_slices)
                assert np.array_equal(bits(interp), bits(want_in[lo:hi])), (rank, mode, b, n_slices)
                assert np.array_equal(order + lo, want_or[lo:hi]), (rank, mode, b, n_slices)
                checked += hi - lo
            # explicit holders (a hash-placed shard hands them over): the same results
            who = (np.searchsorted(bounds, own_ids, side="right") - 1).astype(np.int32)
            ff, interp, _ = scorer.score_owned(qb, queries[q_lo:q_hi], own_off, (own_ids - bounds[who]).astype(np.int32), own_sp, mode, 0.3, n_slices=2, own_holder=who)
            assert np.array_equal(bits(ff), bits(want_ff[lo:hi])) and np.array_equal(bits(interp), bits(want_in[lo:hi]))
            # ... and the candidates as CODES (a numbering all ranks share, here a shuffle of the documents) that only the
            # holder can turn into ordinals: ffx_sharded_set_code_table
            code_of_doc = np.random.default_rng(5).permutation(n_docs).astype(np.int32)
            table = np.full(n_docs + 3, -1, dtype=np.int32)  # (codes nobody uses stay -1)
            table[code_of_doc[d_lo:d_hi]] = np.arange(d_hi - d_lo, dtype=np.int32)
            scorer._s.set_code_table(table)
            ff, interp, order = scorer.score_owned(qb, queries[q_lo:q_hi], own_off, code_of_doc[own_ids], own_sp, mode, 0.3, sort=True, n_slices=3, own_holder=who)
            assert np.array_equal(bits(ff), bits(want_ff[lo:hi])) and np.array_equal(bits(interp), bits(want_in[lo:hi]))
            assert np.array_equal(order + lo, want_or[lo:hi])
            if b == 0 and mode == 2:  # a code the named holder does not hold: the owner is told which pair, everybody raises
                wrong = code_of_doc[own_ids].copy()
                if rank == world - 1 and hi > lo:
                    wrong[hi - lo - 1] = n_docs + 1
                try:
                    scorer.score_owned(qb, queries[q_lo:q_hi], own_off, wrong, own_sp, mode, 0.3, n_slices=1, own_holder=who)
                    raise SystemExit("a code its holder cannot translate was accepted")
                except IndexError as e:
                    assert (f"pair {hi - lo - 1} " in str(e)) if (rank == world - 1 and hi > lo) else ("another rank" in str(e)), str(e)
            scorer._s.set_code_table(None)  # ordinals again
            checked += hi - lo
    # several DIFFERENT batches in flight: stage (i+1) while (i) runs, results of (i-1) leaving
    mode = 2
    outs = [tuple(native.pinned_empty(2000, dt) for dt in (np.float32, np.float32, np.int32)) for _ in range(len(batches))]
    order_of = [0, 1, 2, 1, 0, 2, 2]
    own_off, own_ids, own_sp, _, _ = block(order_of[0])
    scorer.stage_owned(qb, queries[q_lo:q_hi], own_off, own_ids, own_sp, mode)
    results = []
    for i, b in enumerate(order_of):
        scorer.run(0.7, sort=True)
        if i >= 1:
            keep = tuple(np.empty_like(o) for o in outs[0])
            scorer.fetch_async(keep, ff=True, interp=True, order=True)
            results.append((order_of[i - 1], keep))
        if i + 1 < len(order_of):
            own_off, own_ids, own_sp, _, _ = block(order_of[i + 1])
            if i % 2:  # the other way to hand the vectors over: all of them on every rank, nothing exchanged
                scorer.stage_queries(queries)
                scorer.stage_owned(qb, None, own_off, own_ids, own_sp, mode)
                st_h2d = (q_hi - q_lo + 1) * 8 + own_ids.shape[0] * 8 + n_q * d * 4
            else:
                scorer.stage_owned(qb, queries[q_lo:q_hi], own_off, own_ids, own_sp, mode)
                st_h2d = (q_hi - q_lo + 1) * 8 + own_ids.shape[0] * 8 + (q_hi - q_lo) * d * 4
    scorer.sync()
    last = scorer.fetch(order=True)
    results.append((order_of[-1], last))
    for b, (ff, interp, order) in results:
        _, _, _, lo, hi = block(b)
        want_ff, want_in, want_or, _ = expected(b, mode, 0.7)
        m = hi - lo
        assert np.array_equal(bits(ff[:m]), bits(want_ff[lo:hi])) and np.array_equal(bits(interp[:m]), bits(want_in[lo:hi])), (rank, b)
        assert np.array_equal(order[:m] + lo, want_or[lo:hi])
        checked += m
    st = scorer.stats()
    # the intake of a rank is ITS block: offsets, ordinals + first-stage scores, query vectors -- not the batch
    assert st["stage_h2d_bytes"] == st_h2d, (st["stage_h2d_bytes"], st_h2d)
    # an ordinal without rows, in ONE rank's block: the same IndexError on every rank, the position on the rank that owns it
    own_off, own_ids, own_sp, lo, hi = block(0)
    bad_ids = own_ids.copy()
    off0 = batches[0][0]
    bad_rank = next(r for r in range(world) if off0[qb[r + 1]] > off0[qb[r]])  # the first rank whose block holds a candidate (with eight ranks not rank 0)
    if rank == bad_rank:
        bad_ids[min(40, hi - lo - 1)] = int(np.flatnonzero(per_doc == 0)[0])
    try:
        scorer.stage_owned(qb, queries[q_lo:q_hi], own_off, bad_ids, own_sp, mode)
        raise SystemExit("an ordinal without rows was accepted")
    except IndexError as e:
        if rank == bad_rank:
            assert f"pair {min(40, hi - lo - 1)} " in str(e), str(e)
        else:
            assert "another rank" in str(e), str(e)
    if rank == bad_rank:
        bad_ids[min(40, hi - lo - 1)] = n_docs + 5  # outside every shard
    try:
        scorer.stage_owned(qb, queries[q_lo:q_hi], own_off, bad_ids, own_sp, mode)
        raise SystemExit("an ordinal outside every shard was accepted")
    except IndexError:
        pass
    # ... and the scorer is still usable
    ff, interp, _ = scorer.score_owned(qb, queries[q_lo:q_hi], own_off, own_ids, own_sp, mode, 0.3, n_slices=2)
    want_ff, want_in, _, _ = expected(0, mode, 0.3)
    assert np.array_equal(bits(ff), bits(want_ff[lo:hi])) and np.array_equal(bits(interp), bits(want_in[lo:hi]))
    # the set-once form after the fresh form (and back): both leave the other a clean state
    off, ids, sparse = batches[1]
    scorer.set_queries(queries)
    scorer.set_pairs(off, ids, sparse, mode)
    scorer.run(0.3, sort=True)
    ff, interp, _ = scorer.fetch()
    want_ff, want_in, _, _ = expected(1, mode, 0.3)
    lo, hi = int(off[q_lo]), int(off[q_hi])
    assert np.array_equal(bits(ff), bits(want_ff[lo:hi])) and np.array_equal(bits(interp), bits(want_in[lo:hi]))
    own_off, own_ids, own_sp, lo, hi = block(2)
    ff, _, _ = scorer.score_owned(qb, queries[q_lo:q_hi], own_off, own_ids, own_sp, mode, 0.3, n_slices=3)
    assert np.array_equal(bits(ff), bits(expected(2, mode, 0.3)[0][lo:hi]))
    total = int(comm.allreduce([checked])[0])
    if rank == 0:
        print(f"owner-routed world={world}: {total} owned pairs checked bit-exact")
    scorer.close()
    shard.close()
    full.close()
    comm.close()


if __name__ == "__main__":
    import _rank_threads

    _rank_threads.run(main)
