"""Single-rank RCCL run of the native sharded scorer (ffx_comm + ffx_sharded) on one GPU
(launched by tests/test_gpu_parity.py in a fresh interpreter; no torch in the process)."""

import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT / "fast-forward-indexes_amd", ROOT / "oracle", ROOT / "tests"):
    sys.path.insert(0, str(p))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import ffx_oracle as fo  # noqa: E402
from _cases import MODE_NO, bits, load_scoring_case  # noqa: E402
from fast_forward_amd import native  # noqa: E402
from fast_forward_amd.sharded import ShardedScorer, connect, doc_partition  # noqa: E402

case = load_scoring_case("s2_sub")
comm = connect("rccl", rank=0, world=1, device=0)
assert "torch" not in sys.modules
assert comm.allreduce([2.0], native.OP_SUM)[0] == 2.0
comm.barrier()
idx = native.DeviceIndex(case.vectors.shape[0], case.vectors.shape[1])
idx.upload(case.vectors, 0)
idx.set_doc_rows(case.doc_off, case.doc_rows)
idx.set_psg_rows(case.psg_row)
scorer = ShardedScorer(idx, doc_partition(case.doc_off, 1), comm)
scorer.set_queries(case.queries)
pair_off, pair_ids, sparse, _ = case.pairs("MAXP")
scorer.set_pairs(pair_off, pair_ids, sparse, MODE_NO["MAXP"])
for alpha in (0.1, 0.5):
    scorer.run(alpha, exact_sum=True, sort=True)
    scorer.sync()
    ff, interp, order = scorer.fetch(order=True)
    want, want_i = fo.score(case.vectors, case.doc_off, case.doc_rows, case.psg_row, case.queries, pair_off, pair_ids,
                            MODE_NO["MAXP"], sparse, alpha)
    assert np.array_equal(bits(ff), bits(want)) and np.array_equal(bits(interp), bits(want_i))
    assert np.array_equal(order.astype(np.int64), fo.order(want_i, pair_off))
# back-to-back runs alternate between the two plans and overlap scoring with the exchange of the
# run before; the last one wins, and new pairs after that start from a clean state
for alpha in (0.3, 0.7, 0.2, 0.9, 0.05):
    scorer.run(alpha, exact_sum=True, sort=True)
ff, interp, order = scorer.fetch(order=True)
want, want_i = fo.score(case.vectors, case.doc_off, case.doc_rows, case.psg_row, case.queries, pair_off, pair_ids,
                        MODE_NO["MAXP"], sparse, 0.05)
assert np.array_equal(bits(ff), bits(want)) and np.array_equal(bits(interp), bits(want_i))
assert np.array_equal(order.astype(np.int64), fo.order(want_i, pair_off))
pair_off2, pair_ids2, sparse2, _ = case.pairs("FIRSTP")
scorer.set_pairs(pair_off2, pair_ids2, sparse2, MODE_NO["FIRSTP"])
for alpha in (0.6, 0.4, 0.25):
    scorer.run(alpha, exact_sum=True)
ff, interp, _ = scorer.fetch()
want, want_i = fo.score(case.vectors, case.doc_off, case.doc_rows, case.psg_row, case.queries, pair_off2, pair_ids2,
                        MODE_NO["FIRSTP"], sparse2, 0.25)
assert np.array_equal(bits(ff), bits(want)) and np.array_equal(bits(interp), bits(want_i))
scorer.close()
idx.close()

# the Index-API class over the same 1-rank RCCL communicator (control collectives: who holds which
# id, the unknown-id check, gather=True)
import pandas as pd  # noqa: E402

from fast_forward_amd import Ranking  # noqa: E402
from fast_forward_amd.encoder import LambdaEncoder  # noqa: E402
from fast_forward_amd.index import InMemoryIndex, Mode  # noqa: E402
from fast_forward_amd.sharded import ShardedInMemoryIndex  # noqa: E402

rng = np.random.default_rng(3)
n, d = 400, 64
rows = rng.standard_normal((n, d)).astype(np.float32)
qv = rng.standard_normal((4, d)).astype(np.float32)
enc = LambdaEncoder(lambda t: qv[int(t)])
docs = [f"D{i // 2}" for i in range(n)]
sharded, plain = ShardedInMemoryIndex(enc, mode=Mode.MAXP, comm=comm), InMemoryIndex(enc, mode=Mode.MAXP)
sharded.add(rows, doc_ids=docs)
plain.add(rows, doc_ids=docs)
run = {f"q{q}": {f"D{x}": float(rng.random()) for x in rng.choice(n // 2, size=50, replace=False)} for q in range(4)}
first = Ranking.from_run(run, queries={f"q{q}": str(q) for q in range(4)})
assert sharded(first, gather=True) == plain(first)
assert sharded.rerank(first, 0.2, gather=True) == plain.rerank(first, 0.2)
bad = Ranking(pd.concat([first._df, pd.DataFrame([{"q_id": "q0", "id": "nowhere", "score": 1.0, "query": "0"}])]))
try:
    sharded(bad)
    raise SystemExit("an unknown id must raise IndexError")
except IndexError as e:
    assert "nowhere" in str(e)
sharded.close()
comm.close()
assert "torch" not in sys.modules
print("sharded 1-rank ok")
