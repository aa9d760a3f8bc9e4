#!/usr/bin/env python3
"""What this box can stream: gather_dot over every row of an 8.8 M x 768 fp32 index exactly once,
in order and in a random permutation (27.2 GB per launch), next to a device-to-device copy of the
same bytes.  The random gather of the bench should be judged against these, not only against the
8 TB/s data-sheet figure."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "fast-forward-indexes_amd"))
from fast_forward_amd import native  # noqa: E402

n, d = 8_841_823, 768
idx = native.DeviceIndex(n, d)
idx.fill_synthetic(1, 0.03, 0, n)
idx.set_doc_rows(np.arange(n + 1), np.arange(n))
idx.set_psg_rows(np.arange(n))
q = np.ones((4, d), dtype=np.float32)
off = np.array([0, n // 4, n // 2, 3 * (n // 4), n], dtype=np.int64)
plan = native.Plan(idx)
plan.set_queries(q)
out = {}
for label, ids in (("in_order", np.arange(n, dtype=np.int32)), ("permuted", np.random.default_rng(0).permutation(n).astype(np.int32))):
    plan.set_pairs(off, ids, None, 1)
    for exact in (True, False):
        ms = []
        for _ in range(6):
            plan.run(0.0, exact_sum=exact)
            ms.append(plan.stats()["ms_gather_dot"])
        ms = ms[1:]
        out[f"{label}_{'exact' if exact else 'fast'}"] = {"ms": float(np.mean(ms)), "GBs": n * (d * 4 + 4) / np.mean(ms) / 1e6}
dst = native.DeviceIndex(n, d)
dst.copy_rows_from(idx, n)
dst.gather(np.array([n - 1], dtype=np.int32))  # a blocking D2H read: the device-to-device copy above has finished
reps = 4
t0 = time.perf_counter()
for _ in range(reps):
    dst.copy_rows_from(idx, n)  # hipMemcpy device-to-device returns before the copy is done
dst.gather(np.array([n - 1], dtype=np.int32))
t = (time.perf_counter() - t0) / reps
out["memcpy_d2d"] = {"ms": t * 1e3, "GBs_read_plus_write": 2 * n * d * 4 / t / 1e9, "GBs_one_direction": n * d * 4 / t / 1e9}
print(json.dumps(out, indent=1))
