"""One rank of the CPU test of the host-buffer owner routing (``sharded._owner_routed_dense`` and the sharded early stopping built
on it): N plain processes over libffx's shared-memory transport, NO GPU.  The rank's "device" is an ``Index`` whose scoring is the CPU
oracle (``OracleIndex`` below: id tables and call surface of the package, arithmetic of ``oracle/ffx_oracle.c``), so what is
exercised is the protocol -- who is sent what, in which order it comes back, how many collective steps every rank joins, how an
error reaches every rank -- with eight ranks as well as with two.  Launched by tests/test_sharded_native.py."""

import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT / "fast-forward-indexes_amd", ROOT / "oracle", ROOT / "tests"):
    sys.path.insert(0, str(p))

import ffx_oracle as fo  # noqa: E402
from fast_forward_amd import Ranking, native  # noqa: E402
from fast_forward_amd.encoder import LambdaEncoder  # noqa: E402
from fast_forward_amd.index import Mode  # noqa: E402
from fast_forward_amd.index.api import Index  # noqa: E402
from fast_forward_amd.index.ids import IdTable  # noqa: E402
from fast_forward_amd.sharded import _owner_of, _ShardedIndex, connect  # noqa: E402


class OracleIndex(Index):
    """The package's ``Index`` surface with the CPU oracle where the device would be (test double)."""

    def __init__(self, query_encoder, mode):
        super().__init__(query_encoder, mode=mode)
        self._ids = IdTable()
        self._rows = np.zeros((0, 0), dtype=np.float32)

    def _get_internal_dim(self):
        return self._rows.shape[1] if self._rows.size else None

    def _get_doc_ids(self):
        return self._ids.doc_ids

    def _get_psg_ids(self):
        return self._ids.psg_ids

    def _get_num_vectors(self):
        return self._rows.shape[0]

    def _add(self, vectors, doc_ids, psg_ids):
        self._ids.add(list(doc_ids), list(psg_ids), self._rows.shape[0])
        self._rows = vectors.astype(np.float32) if not self._rows.size else np.concatenate([self._rows, vectors.astype(np.float32)])

    def _get_vectors(self, ids):
        raise NotImplementedError

    def _batch_iter(self, batch_size):
        raise NotImplementedError

    def _resolve_ids(self, ids, mode):
        return self._ids.resolve(ids, mode.value)

    def _native_id_table(self, mode):
        return self._ids.native_table(mode.value)

    def _score_pairs(self, query_vectors, pair_off, pair_ids, mode, sparse=None, alpha=None, sort=False, want_dense=True, **_):
        doc_off, doc_rows = self._ids.doc_csr()
        ff, interp = fo.score(self._rows, doc_off, doc_rows, self._ids.psg_rows(), np.asarray(query_vectors, dtype=np.float32), pair_off,
                              np.asarray(pair_ids, dtype=np.int32), mode.value, sparse, 0.0 if alpha is None else alpha)
        key = interp if sparse is not None else ff
        return ff, interp, (fo.order(key, pair_off).astype(np.int32) if sort else None)


class CpuSharded(_ShardedIndex):
    """``_ShardedIndex`` over ``OracleIndex`` shards, rows placed by the hash ``ShardedInMemoryIndex`` uses."""

    def __init__(self, local, comm):
        self.local, self.comm, self.exact_sum = local, comm, True

    def _streams_rows(self):
        return True


comm = connect("host", device=-1)
rank, world = comm.rank, comm.world
rng = np.random.default_rng(7)  # (every rank draws the same collection)
n_docs, d, n_q = 400, 24, 19
per_doc = rng.integers(1, 4, size=n_docs)
doc_names = np.repeat([f"D{i}" for i in range(n_docs)], per_doc)
n_rows = len(doc_names)
psg_names = [f"P{i}" for i in range(n_rows)]
vectors = rng.standard_normal((n_rows, d)).astype(np.float32)
qvecs = rng.standard_normal((n_q, d)).astype(np.float32)
encoder = LambdaEncoder(lambda t: qvecs[int(t)])
lens = [int(x) for x in rng.integers(0, 120, size=n_q)]
lens[3] = 0
run = {f"q{q:02d}": {f"D{c}": float(s) for c, s in zip(rng.choice(n_docs, size=k, replace=False), rng.random(k) * 10)} for q, k in enumerate(lens) if k}
queries = {f"q{q:02d}": str(q) for q in range(n_q)}
first = Ranking.from_run(run, queries={q: queries[q] for q in run})
checked = 0
for mode in (Mode.MAXP, Mode.AVEP, Mode.FIRSTP):
    whole = OracleIndex(encoder, mode)
    whole.add(vectors, doc_ids=list(doc_names), psg_ids=psg_names)
    mine = [i for i in range(n_rows) if _owner_of(doc_names[i], psg_names[i], world) == rank]
    local = OracleIndex(encoder, mode)
    if mine:
        local.add(vectors[mine], doc_ids=[doc_names[i] for i in mine], psg_ids=[psg_names[i] for i in mine])
    sharded = CpuSharded(local, comm)
    # ---- one owner-routed step over the whole batch: this rank's block of the dense scores ------------------------------------
    from fast_forward_amd.index.api import _Run

    run_view = _Run(local, first)
    routing = sharded._code_routing(first)
    qb = native.query_partition(run_view.n_q, world)
    q_lo, q_hi = int(qb[rank]), int(qb[rank + 1])
    off = run_view.pair_off
    lo, hi = int(off[q_lo]), int(off[q_hi])
    rows = slice(lo, hi) if run_view.perm is None else run_view.perm[lo:hi]
    own_codes = np.ascontiguousarray(run_view.codes[rows], dtype=np.int32)
    dense = sharded._owner_routed_dense(run_view.query_vectors, qb, off[q_lo:q_hi + 1] - lo, own_codes, routing)
    want = whole(first)
    want_of = {(q, i): s for q in want.q_ids for i, s in want[q].items()}
    names = run_view.names
    for q in range(q_lo, q_hi):
        for p in range(int(off[q]), int(off[q + 1])):
            got = dense[p - lo]
            assert np.float32(want_of[(run_view.q_order[q], names[own_codes[p - lo]])]) == got, (mode, q, p)
            checked += 1
    it = sharded.last_intake
    held = int((routing["holder_c"][routing["codes"]] == rank).sum())
    assert it == {"own_pairs": hi - lo, "held_pairs": held, "batch_pairs": int(off[-1]),
                  "bytes": (q_hi - q_lo + 1) * 8 + (hi - lo) * 4 + (run_view.n_q + held) * 4}, it
    # ---- early stopping over the shards: every rank runs the loop for its own queries, one collective step per depth --------------
    for cutoff, alpha, depths in ((3, 0.2, (5, 20, 100)), (1, 0.5, (1, 4, 30)), (10, 0.0, (10, 50)), (2, 0.3, (2, 2, 9))):
        want_es = whole(first, early_stopping=cutoff, early_stopping_alpha=alpha, early_stopping_depths=depths)
        assert sharded(first, early_stopping=cutoff, early_stopping_alpha=alpha, early_stopping_depths=depths, gather=True) == want_es, (mode, cutoff)
        own = sharded(first, early_stopping=cutoff, early_stopping_alpha=alpha, early_stopping_depths=depths)
        assert all(own[q] == want_es[q] for q in own.q_ids), (mode, cutoff)
        assert sharded.last_intake["own_pairs"] <= hi - lo and sharded.last_intake["held_pairs"] <= held
    # ---- an id nobody holds: an error on EVERY rank, and only when the loop reaches it -----------------------------------------
    tail = Ranking.from_run({"q00": {"D1": 9.0, "D2": 8.0, "missing-id": 0.5}, "q01": {"D3": 1.0}}, queries={"q00": "0", "q01": "1"})
    assert sharded(tail, early_stopping=1, early_stopping_alpha=1.0, early_stopping_depths=(2, 3), gather=True) == \
        whole(tail, early_stopping=1, early_stopping_alpha=1.0, early_stopping_depths=(2, 3))
    try:
        sharded(tail, early_stopping=3, early_stopping_alpha=0.2, early_stopping_depths=(3, 7), gather=True)
    except IndexError as e:
        assert "missing-id" in str(e), str(e)
    else:
        raise AssertionError("unknown id reached and accepted")
    comm.barrier()  # (every rank raised: the communicator is still in step)
comm.close()
if rank == 0:
    print(f"owner-routed host path, world={world}: {checked} scores of rank 0's block checked bit-exact, early stopping equal to the one-process index")
