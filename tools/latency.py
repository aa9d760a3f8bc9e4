#!/usr/bin/env python3
"""Latency of small batches through the C ABI (a re-ranking service scores one query at a time):
host buffers in -> host buffers out, persistent plan and one-shot ffx_score."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "fast-forward-indexes_amd"))
sys.path.insert(0, str(ROOT))
from fast_forward_amd import native, synth  # noqa: E402
import bench as B  # noqa: E402

rows, d = 8_841_823, 768
scale = float(1 / np.sqrt(d))
idx = native.DeviceIndex(rows, d)
idx.fill_synthetic(2024, scale, 0, rows)
idx.set_doc_rows(np.arange(rows + 1, dtype=np.int64), np.arange(rows, dtype=np.int32))
idx.set_psg_rows(np.arange(rows, dtype=np.int32))
out = {}
for nq, k in ((1, 100), (1, 1000), (1, 5000), (8, 1000), (64, 1000)):
    ids, sparse = B.candidates(rows, nq, k, 7)
    pair_off = np.arange(nq + 1, dtype=np.int64) * k
    q = synth.rows(5, scale, 0, nq, d)
    plan = native.Plan(idx)
    for _ in range(20):
        plan.set_queries(q); plan.set_pairs(pair_off, ids, sparse, 2); plan.run(0.1, sort=True); plan.fetch(interp=True, order=True)
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        plan.set_queries(q); plan.set_pairs(pair_off, ids, sparse, 2); plan.run(0.1, sort=True); plan.fetch(interp=True, order=True)
    t_plan = (time.perf_counter() - t0) / n
    st = plan.stats()
    for _ in range(20):
        native.score(idx, q, pair_off, ids, 2, sparse, 0.1, sort=True)
    t0 = time.perf_counter()
    for _ in range(n):
        native.score(idx, q, pair_off, ids, 2, sparse, 0.1, sort=True)
    t_one = (time.perf_counter() - t0) / n
    out[f"{nq}x{k}"] = {"plan_us": t_plan * 1e6, "ffx_score_us": t_one * 1e6, "device_us": st["ms_total"] * 1e3,
                        "gather_dot_us": st["ms_gather_dot"] * 1e3, "sort_us": st["ms_sort"] * 1e3}
    plan.close()
print(json.dumps(out, indent=1))
