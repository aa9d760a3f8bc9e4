"""N>1 path on CPU: two ranks over gloo run the product's routing + all-to-all exchange and
check every owned query against the single-process oracle (see tests/_gloo_worker.py)."""

import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from fast_forward_amd.sharded import doc_partition, query_partition, route

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_path_over_gloo(world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    port = 29610 + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(ROOT / "tests" / "_gloo_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "checked bit-exact" in out.stdout


def test_partitions_cover_everything_once():
    rng = np.random.default_rng(0)
    per_doc = rng.integers(1, 9, size=1000)
    doc_off = np.concatenate([[0], np.cumsum(per_doc)])
    for world in (1, 2, 4, 8):
        b = doc_partition(doc_off, world)
        assert b[0] == 0 and b[-1] == 1000 and (np.diff(b) >= 0).all()
        rows = np.diff(doc_off[b])
        assert rows.sum() == doc_off[-1] and rows.max() - rows.min() <= 16  # balanced to within two documents
        qb = query_partition(6980, world)
        assert qb[0] == 0 and qb[-1] == 6980 and np.diff(qb).max() - np.diff(qb).min() <= 1


def test_routes_of_all_ranks_are_consistent():
    """What rank s sends to owner o is exactly what o expects from s, pair for pair."""
    rng = np.random.default_rng(1)
    n_docs, n_q, world = 500, 13, 4
    doc_off = np.arange(n_docs + 1)
    lens = rng.integers(0, 40, size=n_q)
    pair_off = np.concatenate([[0], np.cumsum(lens)])
    pair_ids = np.concatenate([rng.choice(n_docs, size=k, replace=False) for k in lens]).astype(np.int32)
    bounds = doc_partition(doc_off, world)
    routes = [route(pair_off, pair_ids, bounds, r, world) for r in range(world)]
    assert sum(r.local_pair_ids.shape[0] for r in routes) == pair_ids.shape[0]
    for o in range(world):
        for s in range(world):
            assert routes[s].send_counts[o] == routes[o].recv_counts[s]
    # simulate the exchange with the global pair position as the payload
    for o in range(world):
        blocks = []
        for s in range(world):
            holder = np.searchsorted(bounds, pair_ids, side="right") - 1
            pair_q = np.repeat(np.arange(n_q), lens)
            pos = np.flatnonzero((holder == s) & (pair_q >= routes[o].q_lo) & (pair_q < routes[o].q_hi))
            blocks.append(pos)
        recv = np.concatenate(blocks)
        lo, hi = pair_off[routes[o].q_lo], pair_off[routes[o].q_hi]
        assert np.array_equal(recv[routes[o].recv_slot], np.arange(lo, hi))
