"""One rank of the world_size-2 CPU rehearsal of the row-sharded path (launched by
tests/test_sharded_gloo.py through torch.distributed.run, backend gloo).

The product's routing (fast_forward_amd.sharded.route / doc_partition, the NumPy statement of
ffx_route) runs unchanged, the exchange is torch.distributed's all_to_all_single over gloo; the two device steps are stood in for by the oracle
(local scoring) and by the one-line NumPy statement of ffx_merge (recv[recv_slot], interpolate).
Every rank checks the queries it owns against the single-process oracle, bit for bit."""

import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT / "fast-forward-indexes_amd", ROOT / "oracle", ROOT / "tests"):
    sys.path.insert(0, str(p))

import ffx_oracle as fo  # noqa: E402
from _cases import MODE_NO, load_scoring_case  # noqa: E402
from fast_forward_amd.sharded import assemble_local, doc_partition, query_partition, route, route_owned  # noqa: E402


def exchange(send, send_counts, recv_counts):
    """The all-to-all-v of dense scores as torch.distributed states it (gloo, CPU tensors): slice j of
    ``send`` goes to rank j.  The product's own transports are ffx_comm_exchange[_host]
    (tests/test_sharded_native.py); this checks the routing against an independent collective."""
    recv = torch.empty(int(np.sum(recv_counts)), dtype=send.dtype)
    dist.all_to_all_single(recv, send, [int(c) for c in recv_counts], [int(c) for c in send_counts])
    return recv


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    checked = 0
    for name, modes in (("config1", ("MAXP", "AVEP", "FIRSTP")), ("d768", ("MAXP",))):
        case = load_scoring_case(name)
        doc_bounds = doc_partition(case.doc_off, world)
        assert doc_bounds[0] == 0 and doc_bounds[-1] == len(case.doc_off) - 1 and (np.diff(doc_bounds) > 0).all()
        d_lo, d_hi = doc_bounds[rank], doc_bounds[rank + 1]
        r_lo, r_hi = case.doc_off[d_lo], case.doc_off[d_hi]
        # this rank's shard: rows of its documents only (fixture docs are stored contiguously)
        shard_vecs = case.vectors[r_lo:r_hi]
        shard_doc_off = case.doc_off[d_lo:d_hi + 1] - r_lo
        shard_rows = np.arange(r_hi - r_lo, dtype=np.int32)
        for mode in modes:
            pair_off, pair_ids, sparse, _ = case.pairs(mode)
            # ragged: drop some candidates so that queries differ in length and one is empty
            keep = np.ones(pair_ids.shape[0], dtype=bool)
            keep[pair_off[1]:pair_off[2]] = False
            keep[pair_off[3]:pair_off[3] + 7] = False
            counts = np.add.reduceat(keep, pair_off[:-1])
            pair_off2 = np.concatenate([[0], np.cumsum(counts)])
            pair_ids2, sparse2 = pair_ids[keep], sparse[keep]
            r = route(pair_off2, pair_ids2, doc_bounds, rank, world, sparse2)
            assert r.send_counts.sum() == r.local_pair_ids.shape[0]
            local_ff, _ = fo.score(shard_vecs, shard_doc_off, shard_rows, shard_rows, case.queries, r.local_pair_off,
                                   r.local_pair_ids, MODE_NO[mode])
            recv = exchange(torch.from_numpy(local_ff), r.send_counts, r.recv_counts).numpy()
            dense = recv[r.recv_slot]                                  # ffx_merge: scatter back to first-stage order
            interp = fo.interpolate(r.own_sparse, dense, 0.2)          # ... and interpolate
            want_ff, want_i = fo.score(case.vectors, case.doc_off, case.doc_rows, case.psg_row, case.queries, pair_off2,
                                       pair_ids2, MODE_NO[mode], sparse2, 0.2)
            lo, hi = pair_off2[r.q_lo], pair_off2[r.q_hi]
            assert lo == r.own_first_pair
            assert np.array_equal(dense.view(np.uint32), want_ff[lo:hi].view(np.uint32)), (name, mode, rank)
            assert np.array_equal(interp.view(np.uint32), want_i[lo:hi].view(np.uint32)), (name, mode, rank)
            checked += hi - lo
            # the same batch handed over the way a serving loop would: every rank sees ONLY the queries it owns
            # (ffx_sharded_stage_owned).  Message A: per-query counts to the holders; message B: the ordinals.
            qb = query_partition(len(pair_off2) - 1, world)
            own = route_owned(pair_off2[qb[rank]:qb[rank + 1] + 1] - lo, pair_ids2[lo:hi], rank, world, doc_bounds=doc_bounds)
            n_own_q = int(qb[rank + 1] - qb[rank])
            cnt = exchange(torch.from_numpy(own.counts.reshape(-1)), [n_own_q] * world, np.diff(qb)).numpy()
            ids_in = exchange(torch.from_numpy(own.send_ids), np.diff(own.starts), np.add.reduceat(cnt, qb[:-1]) if cnt.size else np.zeros(world, dtype=np.int64)).numpy() \
                if all(qb[i + 1] > qb[i] for i in range(world)) else None
            if ids_in is not None:
                got_by = np.add.reduceat(cnt, qb[:-1])
                bounds_in = np.concatenate([[0], np.cumsum(got_by)])
                l_off, l_ids, send_back = assemble_local(qb, [cnt[qb[o]:qb[o + 1]] for o in range(world)], [ids_in[bounds_in[o]:bounds_in[o + 1]] for o in range(world)])
                assert np.array_equal(l_off, r.local_pair_off) and np.array_equal(l_ids, r.local_pair_ids) and np.array_equal(send_back, r.send_counts)
                assert np.array_equal(own.slot, r.recv_slot) and np.array_equal(np.diff(own.starts), r.recv_counts)
                recv2 = exchange(torch.from_numpy(local_ff), send_back, np.diff(own.starts)).numpy()
                assert np.array_equal(recv2[own.slot].view(np.uint32), want_ff[lo:hi].view(np.uint32)), (name, mode, rank, "owner-routed")
        # every query is owned exactly once
        qb = query_partition(len(case.q_names), world)
        assert qb[0] == 0 and qb[-1] == len(case.q_names)
    total = torch.tensor([checked])
    dist.all_reduce(total)
    if rank == 0:
        print(f"gloo world={world}: {int(total)} owned pairs checked bit-exact")
    # an ordinal outside the sharded index is the reference's IndexError, raised while routing
    try:
        route(np.array([0, 1]), np.array([10**6]), np.array([0, 5, 10]), rank, world)
    except IndexError:
        pass
    else:
        raise AssertionError("route accepted an unknown ordinal")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
