#!/usr/bin/env python3
"""Does the ORDER of a query's candidates matter to gather_dot?  Same 8.8 M x 768 index, same 34.9 M
(query, row) pairs; candidate lists (a) in first-stage order (random rows), (b) sorted by row inside every
query, (c) the whole batch's rows in ascending order through 4 pseudo-queries (the in-order ceiling)."""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "fast-forward-indexes_amd"))
sys.path.insert(0, str(ROOT))
from fast_forward_amd import native, synth  # noqa: E402
import bench as B  # noqa: E402

n, d, nq, k = 8_841_823, 768, 6980, 5000
idx = native.DeviceIndex(n, d)
idx.fill_synthetic(2024, 0.036, 0, n)
idx.set_doc_rows(np.arange(n + 1, dtype=np.int64), np.arange(n, dtype=np.int32))
idx.set_psg_rows(np.arange(n, dtype=np.int32))
ids, sparse = B.candidates(n, nq, k, 2024)
pair_off = np.arange(nq + 1, dtype=np.int64) * k
plan = native.Plan(idx)
plan.set_queries(synth.rows(2027, 0.036, 0, nq, d))
out = {}
for label, cand in (("first_stage_order", ids), ("sorted_by_row_inside_each_query", np.sort(ids.reshape(nq, k), axis=1).reshape(-1))):
    plan.set_pairs(pair_off, cand, sparse, 1)
    ms = []
    for _ in range(8):
        plan.run(0.1, exact_sum=True)
        ms.append(plan.stats()["ms_gather_dot"])
    out[label] = {"ms": float(np.mean(ms[2:])), "GBs": nq * k * (d * 4 + 4) / float(np.mean(ms[2:])) / 1e6}
print(json.dumps(out, indent=1))
