#!/usr/bin/env python3
"""Extreme batch shapes through one plan (many tiny queries, a few huge ones): stage times and a sortedness check."""
import sys, time, json
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "fast-forward-indexes_amd")); sys.path.insert(0, str(ROOT))
import numpy as np
from fast_forward_amd import native, synth
import bench as B
rows, d = 8_841_823, 768
scale = float(1 / np.sqrt(d))
idx = native.DeviceIndex(rows, d); idx.fill_synthetic(2024, scale, 0, rows)
idx.set_doc_rows(np.arange(rows + 1, dtype=np.int64), np.arange(rows, dtype=np.int32)); idx.set_psg_rows(np.arange(rows, dtype=np.int32))
for nq, k in ((100_000, 100), (1, 5_000_000), (20, 500_000), (1_000_000, 10), (100, 60_000), (200, 30_000), (400, 15_000)):
    ids, sparse = B.candidates(rows, nq, k, 7); pair_off = np.arange(nq + 1, dtype=np.int64) * k; q = synth.rows(5, scale, 0, nq, d)
    plan = native.Plan(idx); plan.set_queries(q); plan.set_pairs(pair_off, ids, sparse, 2)
    plan.run(0.1, sort=True); plan.sync()
    t0 = time.perf_counter(); plan.run(0.1, sort=True); plan.sync(); dt = time.perf_counter() - t0
    st = plan.stats()
    ff, interp, order = plan.fetch(interp=True, order=True)
    o = order[: k].astype(np.int64); ok = bool((np.diff(interp[o]) <= 0).all())
    print(nq, k, "ms", round(dt * 1e3, 2), {kk: round(st[kk], 3) for kk in ("ms_expand", "ms_gather_dot", "ms_reduce", "ms_sort")}, "sorted", ok, flush=True)
    plan.close()
