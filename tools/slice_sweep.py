#!/usr/bin/env python3
"""Region (ii) at the headline size for several (slice, ramp) sizes of ffx_score's pipeline (FFX_DEBUG_VARIANT=score_slice_pairs=N,
score_ramp_pairs=N): page-locked buffers (argv[1] = "ordinary": ordinary memory both ways), best of 5 calls each, the
HBM-resident step beside it."""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "fast-forward-indexes_amd"))
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import _experiments  # noqa: E402

_experiments.use()  # the switches this tool flips exist in the experiments build only
from fast_forward_amd import native, synth  # noqa: E402
import bench as B  # noqa: E402

rows, d, nq, k = 8_841_823, 768, 6980, 5000
scale = float(1 / np.sqrt(d))
idx = native.DeviceIndex(rows, d)
idx.fill_synthetic(2024, scale, 0, rows)
idx.set_doc_rows(np.arange(rows + 1, dtype=np.int64), np.arange(rows, dtype=np.int32))
idx.set_psg_rows(np.arange(rows, dtype=np.int32))
ids, sparse = B.candidates(rows, nq, k, 2024)
pair_off = np.arange(nq + 1, dtype=np.int64) * k
queries = synth.rows(2027, scale, 0, nq, d)
n = int(pair_off[-1])
p_q = native.pinned_empty(queries.shape, np.float32)
p_ids, p_sp = native.pinned_empty(n, np.int32), native.pinned_empty(n, np.float32)
p_q[...], p_ids[...], p_sp[...] = queries, ids, sparse
outs = (native.pinned_empty(n, np.float32), native.pinned_empty(n, np.float32), native.pinned_empty(n, np.int32))
if len(sys.argv) > 1 and sys.argv[1] == "ordinary":
    p_q, p_ids, p_sp = queries, ids, sparse
    outs = (np.zeros(n, dtype=np.float32), np.zeros(n, dtype=np.float32), np.zeros(n, dtype=np.int32))
run = lambda: native.score(idx, p_q, pair_off, p_ids, 2, p_sp, 0.1, exact_sum=True, sort=True, out=outs)  # noqa: E731
plan = native.Plan(idx)
plan.set_queries(queries)
plan.set_pairs(pair_off, ids, sparse, 2)
for _ in range(3):
    plan.run(0.1, exact_sum=True, sort=True)
    plan.sync()
t0 = time.perf_counter()
for _ in range(5):
    plan.run(0.1, exact_sum=True, sort=True)
plan.sync()
out = {"resident_step_ms": (time.perf_counter() - t0) / 5 * 1e3}
want = None
for slice_pairs, ramp in ((0, 0), (4 << 20, 256 << 10), (6 << 20, 512 << 10), (8 << 20, 512 << 10), (8 << 20, 256 << 10), (12 << 20, 512 << 10),
                          (16 << 20, 512 << 10), (6 << 20, 768 << 10), (3 << 20, 384 << 10), (6 << 20, 128 << 10), (4 << 20, 128 << 10), (3 << 20, 128 << 10),
                          (6 << 20, 384 << 10), (5 << 20, 256 << 10)):
    words = ([f"score_slice_pairs={slice_pairs}"] if slice_pairs else []) + ([f"score_ramp_pairs={ramp}"] if ramp else [])
    if words:
        os.environ["FFX_DEBUG_VARIANT"] = ",".join(words)
    else:
        os.environ.pop("FFX_DEBUG_VARIANT", None)
    run()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        res = run()
        ts.append(time.perf_counter() - t0)
    if want is None:
        want = tuple(a.copy() for a in res[:3])
    same = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(res[:3], want))
    out[f"slice={slice_pairs >> 10}K ramp={ramp >> 10}K"] = {"best_ms": min(ts) * 1e3, "median_ms": float(np.median(ts)) * 1e3, "same_bits": bool(same)}
print(json.dumps(out, indent=1))
