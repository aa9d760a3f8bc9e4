"""The multi-GPU half of the C ABI on CPU: native routing against its NumPy statement, and N plain
processes talking through libffx's own communicator (shared-memory transport; no GPU, no torch)."""

import os
import subprocess
import sys
import uuid
from pathlib import Path

import numpy as np
import pytest

from fast_forward_amd import native
from fast_forward_amd.sharded import assemble_local, doc_partition, query_partition, route, route_owned

ROOT = Path(__file__).resolve().parent.parent


def run_ranks(world: int, script: Path, *args, timeout: int = 600, extra_env=None, ranks_per_process: int = 1):
    """Start the ranks of one job and wait for them: (exit codes, (stdout, stderr)) per PROCESS.  ``ranks_per_process`` > 1: a
    process hosts that many consecutive ranks, a thread each (FFX_RANKS_HERE names them; a worker that supports it starts the
    threads itself) -- how eight ranks share one GPU on a box that admits six processes to it."""
    key = f"test_{os.getpid()}_{uuid.uuid4().hex[:10]}"
    procs = []
    for r in range(0, world, ranks_per_process):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), FFX_RDZV_KEY=key, OMP_NUM_THREADS="2",
                   FFX_COMM_TIMEOUT_S="60", **(extra_env or {}))
        if ranks_per_process > 1:
            env["FFX_RANKS_HERE"] = ",".join(str(x) for x in range(r, min(world, r + ranks_per_process)))
        procs.append(subprocess.Popen([sys.executable, str(script), *args], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=timeout))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    return [p.returncode for p in procs], outs


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_path_over_native_host_transport(world):
    codes, outs = run_ranks(world, ROOT / "tests" / "_host_comm_worker.py")
    assert codes == [0] * world, "\n".join(o[0][-1500:] + o[1][-3000:] for o in outs)
    assert "checked bit-exact" in outs[0][0]


@pytest.mark.parametrize("world", [2, 3, 8])
def test_owner_routed_host_path_and_sharded_early_stopping_without_a_gpu(world):
    """``sharded._owner_routed_dense`` (streamed shards, the steps of early stopping) and ``_ShardedIndex._early_stopping`` with 2, 3
    and EIGHT ranks over the shared-memory transport, the CPU oracle standing in for each rank's device
    (tests/_owner_routed_cpu_worker.py): every rank's block of dense scores bit-equal to the one-process index's, per-rank intake =
    its own block + what it holds, early stopping equal to the one-process loop for four (cutoff, depths) settings in three modes, and
    an id nobody holds raised on every rank at the same collective step whichever rank's loop reaches it."""
    codes, outs = run_ranks(world, ROOT / "tests" / "_owner_routed_cpu_worker.py", timeout=600)
    assert codes == [0] * world, "\n".join(o[0][-1500:] + o[1][-3000:] for o in outs)
    assert f"world={world}:" in outs[0][0] and "checked bit-exact" in outs[0][0]


def test_ranks_started_by_torchrun_find_each_other():
    """The other way N ranks get started (`python -m torch.distributed.run ...`): no FFX_RDZV_KEY; the ranks
    derive the job's name from the launcher's port, run id and their common parent process."""
    env = {k: v for k, v in os.environ.items() if k not in ("FFX_RDZV_KEY", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(OMP_NUM_THREADS="2", FFX_COMM_TIMEOUT_S="60")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29644", str(ROOT / "tests" / "_host_comm_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "host transport world=2" in out.stdout and "checked bit-exact" in out.stdout


def test_native_query_partition_matches_numpy():
    for n_q in (0, 1, 7, 6980, 10**6 + 3):
        for world in (1, 2, 3, 8):
            assert np.array_equal(native.query_partition(n_q, world), query_partition(n_q, world))


@pytest.mark.parametrize("seed", range(4))
def test_native_route_equals_numpy_route(seed):
    rng = np.random.default_rng(seed)
    n_docs, n_q = int(rng.integers(50, 3000)), int(rng.integers(1, 40))
    per_doc = rng.integers(1, 6, size=n_docs)
    doc_off = np.concatenate([[0], np.cumsum(per_doc)])
    lens = rng.integers(0, min(n_docs, 200), size=n_q)
    pair_off = np.concatenate([[0], np.cumsum(lens)])
    pair_ids = np.concatenate([rng.choice(n_docs, size=k, replace=False) for k in lens] + [np.zeros(0, dtype=np.int64)]).astype(np.int32)
    sparse = rng.random(pair_ids.shape[0]).astype(np.float32)
    for world in (1, 2, 3, 5, 8):
        bounds = doc_partition(doc_off, world)
        for rank in range(world):
            want = route(pair_off, pair_ids, bounds, rank, world, sparse)
            got = native.route(pair_off, pair_ids, rank, world, doc_bounds=bounds)
            assert np.array_equal(got["local_pair_off"], want.local_pair_off)
            assert np.array_equal(got["local_pair_ids"], want.local_pair_ids)
            assert np.array_equal(got["send_counts"], want.send_counts) and np.array_equal(got["recv_counts"], want.recv_counts)
            assert np.array_equal(got["recv_slot"], want.recv_slot)
            assert (got["q_lo"], got["q_hi"], got["own_first_pair"]) == (want.q_lo, want.q_hi, want.own_first_pair)
            # explicit holders (ids already local to their holder) route the same way
            holder = (np.searchsorted(bounds, pair_ids, side="right") - 1).astype(np.int32)
            local = (pair_ids - bounds[holder]).astype(np.int32)
            held = native.route(pair_off, local, rank, world, holder=holder)
            for key in ("local_pair_off", "local_pair_ids", "send_counts", "recv_counts", "recv_slot"):
                assert np.array_equal(held[key], got[key]), key


@pytest.mark.parametrize("seed", range(6))
def test_owner_side_routing_delivers_what_full_routing_derives(seed):
    """ffx_sharded_stage_owned's protocol (its NumPy statement): every rank is handed only the block of queries it OWNS,
    groups its candidates by holder (message A: per-query counts, message B: holder-local ordinals) and the messages of
    the owners in rank order ARE each holder's local CSR; the owner-side arrival slots are the ones the full routing
    derives from the whole batch.  Document ranges and explicit holders, ragged and empty queries, empty blocks."""
    rng = np.random.default_rng(100 + seed)
    n_docs, n_q = int(rng.integers(50, 3000)), int(rng.integers(1, 40))
    per_doc = rng.integers(1, 6, size=n_docs)
    doc_off = np.concatenate([[0], np.cumsum(per_doc)])
    lens = rng.integers(0, min(n_docs, 200), size=n_q)
    lens[rng.integers(0, n_q)] = 0
    pair_off = np.concatenate([[0], np.cumsum(lens)])
    pair_ids = np.concatenate([rng.choice(n_docs, size=k, replace=False) for k in lens] + [np.zeros(0, dtype=np.int64)]).astype(np.int32)
    for world in (1, 2, 3, 5, 8):
        bounds = doc_partition(doc_off, world)
        qb = query_partition(n_q, world)
        who = (np.searchsorted(bounds, pair_ids, side="right") - 1).astype(np.int32)
        for explicit in (False, True):
            owned = []
            for r in range(world):
                lo, hi = pair_off[qb[r]], pair_off[qb[r + 1]]
                if explicit:
                    owned.append(route_owned(pair_off[qb[r]:qb[r + 1] + 1] - lo, (pair_ids - bounds[who])[lo:hi], r, world, holder=who[lo:hi]))
                else:
                    owned.append(route_owned(pair_off[qb[r]:qb[r + 1] + 1] - lo, pair_ids[lo:hi], r, world, doc_bounds=bounds))
            for h in range(world):
                want = route(pair_off, pair_ids, bounds, h, world)
                l_off, l_ids, back = assemble_local(qb, [owned[o].counts[h] for o in range(world)],
                                                    [owned[o].send_ids[owned[o].starts[h]:owned[o].starts[h + 1]] for o in range(world)])
                assert np.array_equal(l_off, want.local_pair_off) and np.array_equal(l_ids, want.local_pair_ids) and np.array_equal(back, want.send_counts)
                assert np.array_equal(np.diff(owned[h].starts), want.recv_counts) and np.array_equal(owned[h].slot, want.recv_slot)
    with pytest.raises(IndexError):
        route_owned(np.array([0, 1]), np.array([10]), 0, 2, doc_bounds=np.array([0, 5, 10]))


def test_native_route_large_batch_uses_threads_and_stays_exact():
    rng = np.random.default_rng(7)
    n_docs, n_q, k = 200_000, 700, 1500  # > 2^18 pairs per thread block: several threads
    pair_off = np.arange(n_q + 1, dtype=np.int64) * k
    pair_ids = rng.integers(0, n_docs, size=n_q * k).astype(np.int32)
    bounds = doc_partition(np.arange(n_docs + 1), 4)
    for rank in (0, 3):
        want = route(pair_off, pair_ids, bounds, rank, 4)
        got = native.route(pair_off, pair_ids, rank, 4, doc_bounds=bounds)
        assert np.array_equal(got["local_pair_ids"], want.local_pair_ids) and np.array_equal(got["recv_slot"], want.recv_slot)


def test_native_route_rejects_ordinals_outside_every_shard():
    with pytest.raises(IndexError, match="pair 1 names document ordinal 10"):
        native.route(np.array([0, 2]), np.array([3, 10], dtype=np.int32), 0, 2, doc_bounds=np.array([0, 5, 10]))
    with pytest.raises(IndexError):
        native.route(np.array([0, 1]), np.array([-1], dtype=np.int32), 1, 2, doc_bounds=np.array([0, 5, 10]))
    with pytest.raises(IndexError):  # a holder that is no rank
        native.route(np.array([0, 1]), np.array([0], dtype=np.int32), 0, 2, holder=np.array([2], dtype=np.int32))


def test_single_rank_communicator_needs_no_peer():
    comm = native.Comm(native.COMM_HOST, -1, 0, 1, native.Comm.unique_id(native.COMM_HOST))
    comm.barrier()
    assert comm.allreduce([3.0], native.OP_MAX)[0] == 3.0
    out = comm.exchange_host(np.arange(5, dtype=np.float32), [5], [5])
    assert np.array_equal(out, np.arange(5, dtype=np.float32))
    comm.close()


def test_rccl_transport_without_a_gpu_fails_loudly():
    if native.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises((native.FFXError, RuntimeError)):
        native.Comm(native.COMM_RCCL, 0, 0, 1, b"\0" * native.COMM_ID_BYTES)
