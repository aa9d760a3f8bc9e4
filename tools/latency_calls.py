#!/usr/bin/env python3
"""Where the 0.19 ms of a single small query (1 x k=1000, host buffers in and out) go, call by call."""
import sys, time, json
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "fast-forward-indexes_amd")); sys.path.insert(0, str(ROOT))
import numpy as np
from fast_forward_amd import native, synth
import bench as B
rows, d = 2_000_000, 768
scale = float(1 / np.sqrt(d))
idx = native.DeviceIndex(rows, d); idx.fill_synthetic(2024, scale, 0, rows)
idx.set_doc_rows(np.arange(rows + 1, dtype=np.int64), np.arange(rows, dtype=np.int32)); idx.set_psg_rows(np.arange(rows, dtype=np.int32))
nq, k = 1, 1000
ids, sparse = B.candidates(rows, nq, k, 7); pair_off = np.arange(nq + 1, dtype=np.int64) * k; q = synth.rows(5, scale, 0, nq, d)
plan = native.Plan(idx)
T = {"set_queries": 0.0, "set_pairs": 0.0, "run": 0.0, "sync": 0.0, "fetch": 0.0}
n = 500
for it in range(n + 50):
    t0 = time.perf_counter(); plan.set_queries(q)
    t1 = time.perf_counter(); plan.set_pairs(pair_off, ids, sparse, 2)
    t2 = time.perf_counter(); plan.run(0.1, sort=True)
    t3 = time.perf_counter(); plan.sync()
    t4 = time.perf_counter(); plan.fetch(interp=True, order=True)
    t5 = time.perf_counter()
    if it >= 50:
        for key, dt in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): T[key] += dt
print({k2: round(v / n * 1e6, 1) for k2, v in T.items()}, plan.stats()["ms_total"] * 1e3)
lib = native.lib()
t0 = time.perf_counter()
for _ in range(2000): lib.ffx_abi_version()
print("ctypes call us", (time.perf_counter() - t0) / 2000 * 1e6)
