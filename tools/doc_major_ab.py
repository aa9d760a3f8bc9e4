#!/usr/bin/env python3
"""Experiment (experiments build): does fetching every index row ONCE for all the queries that name it pay?  The headline batch
(6980 x 5000 candidates over 8.84 M rows: every row is named 3.95 times on average), HBM-resident step, variants of FFX_DEBUG_VARIANT:
  doc_major             the row list sorted by index row before gather_dot (candidates of one row adjacent: the row comes from HBM once,
                        the others find it in the cache); the query vector then changes with every row (21 MB of queries: beyond one L2)
  seg_leaves=N          rows scored in pieces of N leaves of NumPy's tree (d = 768, N = 1: eight passes over 96-element slices; the query
                        slices of a pass are 2.7 MB in all and stay in every L2), partial sums combined in NumPy's order: same bits
Stages from the plan's HIP events; scores compared bit for bit with the shipping path."""
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

_experiments.use()
from fast_forward_amd import native, synth  # noqa: E402
import bench as B  # noqa: E402

rows, d, nq, k = 8_841_823, 768, int(sys.argv[1]) if len(sys.argv) > 1 else 6980, 5000
scale = float(1 / np.sqrt(d))
idx = native.DeviceIndex(rows, d)
idx.fill_synthetic(2024, scale, 0, rows)
idx.set_doc_rows(np.arange(rows + 1, dtype=np.int64), np.arange(rows, dtype=np.int32))
idx.set_psg_rows(np.arange(rows, dtype=np.int32))
ids, sparse = B.candidates(rows, nq, k, 2024)
pair_off = np.arange(nq + 1, dtype=np.int64) * k
queries = synth.rows(2027, scale, 0, nq, d)
out, want = {}, None
for label, variant in (("shipping", ""), ("doc_major", "doc_major"), ("seg_leaves=1", "seg_leaves=1"), ("doc_major + seg_leaves=1", "doc_major,seg_leaves=1"),
                       ("doc_major + seg_leaves=2", "doc_major,seg_leaves=2"), ("doc_major + seg_leaves=4", "doc_major,seg_leaves=4")):
    if variant:
        os.environ["FFX_DEBUG_VARIANT"] = variant
    else:
        os.environ.pop("FFX_DEBUG_VARIANT", None)
    plan = native.Plan(idx)  # (the pieces are decided when the plan is made)
    plan.set_queries(queries)
    plan.set_pairs(pair_off, ids, sparse, 2)
    for _ in range(2):
        plan.run(0.1, exact_sum=True, sort=True)
        plan.sync()
    steps, stages = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        plan.run(0.1, exact_sum=True, sort=True)
        plan.sync()
        steps.append((time.perf_counter() - t0) * 1e3)
        stages.append(plan.stats())
    ff, interp, order = plan.fetch(interp=True, order=True)
    if want is None:
        want = (ff.copy(), interp.copy(), order.copy())
    same = bool(np.array_equal(ff.view(np.uint32), want[0].view(np.uint32)) and np.array_equal(interp.view(np.uint32), want[1].view(np.uint32)) and np.array_equal(order, want[2]))
    best = min(range(5), key=lambda i: steps[i])
    st = stages[best]
    out[label] = {"step_ms": round(steps[best], 3), "ms_expand": round(st["ms_expand"], 3), "ms_gather_dot": round(st["ms_gather_dot"], 3),
                  "ms_reduce": round(st["ms_reduce"], 3), "ms_sort": round(st["ms_sort"], 3), "kernel_variant": st["kernel_variant"], "same_bits": same}
    print(label, out[label], flush=True)
    plan.close()
print(json.dumps(out))
