"""One rank of the CPU rehearsal of the row-sharded path over libffx's OWN communicator (shared-memory
transport, no GPU, no torch): launched by tests/test_sharded_native.py as N plain processes.

Native routing (ffx_route) + native all-to-all-v (ffx_comm_exchange_host) run unchanged; the two
device steps are stood in for by the oracle (local scoring) and by the one-line NumPy statement
of ffx_merge (recv[recv_slot], interpolate).  Every rank checks the queries it owns against the
single-process oracle, bit for bit."""

import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT / "fast-forward-indexes_amd", ROOT / "oracle", ROOT / "tests"):
    sys.path.insert(0, str(p))

import ffx_oracle as fo  # noqa: E402
from _cases import MODE_NO, load_scoring_case  # noqa: E402
from fast_forward_amd import native  # noqa: E402
from fast_forward_amd.sharded import assemble_local, connect, doc_partition, query_partition, route_owned  # noqa: E402


def main():
    comm = connect("host", device=-1)
    rank, world = comm.rank, comm.world
    assert world == int(os.environ["WORLD_SIZE"]) and rank == int(os.environ["RANK"])
    # control plane
    assert comm.allreduce([rank + 1.0, -rank], native.OP_SUM).tolist() == [world * (world + 1) / 2, -world * (world - 1) / 2]
    assert comm.allreduce([float(rank)], native.OP_MAX)[0] == world - 1 and comm.allreduce([float(rank)], native.OP_MIN)[0] == 0
    big = np.full(100_000, rank, dtype=np.int32)  # several control rounds
    got = comm.allgather_array(big)
    assert got.shape == (world, 100_000) and all((got[r] == r).all() for r in range(world))
    assert comm.allgather_bytes(bytes([65 + rank]) * 3) == [bytes([65 + r]) * 3 for r in range(world)]
    comm.barrier()
    checked = 0
    for name, modes in (("config1", ("MAXP", "AVEP", "FIRSTP")), ("d768", ("MAXP",))):
        case = load_scoring_case(name)
        doc_bounds = doc_partition(case.doc_off, world)
        d_lo, d_hi = doc_bounds[rank], doc_bounds[rank + 1]
        r_lo, r_hi = case.doc_off[d_lo], case.doc_off[d_hi]
        shard_vecs = case.vectors[r_lo:r_hi]
        shard_doc_off = case.doc_off[d_lo:d_hi + 1] - r_lo
        shard_rows = np.arange(r_hi - r_lo, dtype=np.int32)
        for mode in modes:
            pair_off, pair_ids, sparse, _ = case.pairs(mode)
            keep = np.ones(pair_ids.shape[0], dtype=bool)  # ragged: queries differ in length, one is empty
            keep[pair_off[1]:pair_off[2]] = False
            keep[pair_off[3]:pair_off[3] + 7] = False
            counts = np.add.reduceat(keep, pair_off[:-1])
            pair_off2 = np.concatenate([[0], np.cumsum(counts)])
            pair_ids2, sparse2 = pair_ids[keep], sparse[keep]
            r = native.route(pair_off2, pair_ids2, rank, world, doc_bounds=doc_bounds)
            local_ff, _ = fo.score(shard_vecs, shard_doc_off, shard_rows, shard_rows, case.queries, r["local_pair_off"],
                                   r["local_pair_ids"], MODE_NO[mode])
            for _ in range(2):  # twice: the second exchange reuses the segments
                recv = comm.exchange_host(local_ff, r["send_counts"], r["recv_counts"])
            dense = recv[r["recv_slot"]]
            lo = r["own_first_pair"]
            hi = lo + r["own_pairs"]
            interp = fo.interpolate(sparse2[lo:hi], dense, 0.2)
            want_ff, want_i = fo.score(case.vectors, case.doc_off, case.doc_rows, case.psg_row, case.queries, pair_off2,
                                       pair_ids2, MODE_NO[mode], sparse2, 0.2)
            assert np.array_equal(dense.view(np.uint32), want_ff[lo:hi].view(np.uint32)), (name, mode, rank)
            assert np.array_equal(interp.view(np.uint32), want_i[lo:hi].view(np.uint32)), (name, mode, rank)
            checked += hi - lo
            # the same batch with every rank handed ONLY the queries it owns (the protocol of ffx_sharded_stage_owned over the
            # native all-to-all-v: words travel as bit patterns): counts to the holders, ordinals to the holders, scores back
            qb = query_partition(len(pair_off2) - 1, world)
            own = route_owned(pair_off2[qb[rank]:qb[rank + 1] + 1] - lo, pair_ids2[lo:hi], rank, world, doc_bounds=doc_bounds)
            n_own_q = int(qb[rank + 1] - qb[rank])
            cnt = comm.exchange_host(own.counts.reshape(-1).view(np.float32), [n_own_q] * world, np.diff(qb)).view(np.int32)
            got_by = np.array([int(cnt[qb[o]:qb[o + 1]].sum()) for o in range(world)], dtype=np.int64)
            ids_in = comm.exchange_host(own.send_ids.view(np.float32), np.diff(own.starts), got_by).view(np.int32)
            at = np.concatenate([[0], np.cumsum(got_by)])
            l_off, l_ids, send_back = assemble_local(qb, [cnt[qb[o]:qb[o + 1]] for o in range(world)], [ids_in[at[o]:at[o + 1]] for o in range(world)])
            assert np.array_equal(l_off, r["local_pair_off"]) and np.array_equal(l_ids, r["local_pair_ids"]) and np.array_equal(send_back, r["send_counts"])
            assert np.array_equal(own.slot, r["recv_slot"])
            recv2 = comm.exchange_host(local_ff, send_back, np.diff(own.starts))
            assert np.array_equal(recv2[own.slot].view(np.uint32), want_ff[lo:hi].view(np.uint32)), (name, mode, rank, "owner-routed")
    total = int(comm.allreduce([checked])[0])
    # a growing payload re-creates the data segments (new generation): peers must re-map
    for n in (10, 300_000, 5, 2_000_000):
        send = np.arange(n * world, dtype=np.float32) + rank * 1e7
        recv = comm.exchange_host(send, [n] * world, [n] * world)
        for src in range(world):
            assert np.array_equal(recv[src * n:(src + 1) * n], np.arange(rank * n, (rank + 1) * n, dtype=np.float32) + np.float32(src * 1e7))
    # counts that do not match across ranks are an error on every rank, not a hang
    try:
        comm.exchange_host(np.zeros(world, dtype=np.float32), [1] * world, [1 if rank == 0 else 2] * world)
    except native.FFXError as e:
        assert e.code == native.RCCL_ERROR or e.code == native.BAD_ARG
    else:
        if world > 1:
            raise AssertionError("mismatched counts were accepted")
    if rank == 0:
        print(f"host transport world={world}: {total} owned pairs checked bit-exact")
    comm.close()


if __name__ == "__main__":
    main()
