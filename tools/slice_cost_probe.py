#!/usr/bin/env python3
"""Why do the kernels of ffx_score's slices take longer in sum (17.8 ms) than one launch over the whole batch (16.6 ms)?
The same slices, already set on plans of their own, are run back to back on ONE stream with nothing else going on: if
their sum is the sliced figure, the cost is in cutting a launch into 14 (tails / ramps of every launch); if it is the
one-launch figure, the cost is what runs beside the kernels inside ffx_score (uploads, validation kernels, downloads)."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "fast-forward-indexes_amd"))
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from fast_forward_amd import native  # noqa: E402

rows, dim, nq, k = 8_841_823, 768, 6980, 5000
index = native.DeviceIndex(rows, dim)
index.fill_synthetic(7, 0.036, 0, rows)
index.set_doc_rows(np.arange(rows + 1, dtype=np.int64), np.arange(rows, dtype=np.int32))
index.set_psg_rows(np.arange(rows, dtype=np.int32))
ids, sparse = bench.candidates(rows, nq, k, 2024)
qv = (np.random.default_rng(1).standard_normal((nq, dim)) * 0.036).astype(np.float32)
out = {}


def timed(plans, reps=5):
    best = 1e9
    for _ in range(reps):
        for p in plans:
            p.sync()
        t0 = time.perf_counter()
        for p in plans:
            p.run(0.1, exact_sum=True, sort=True)
        for p in plans:
            p.sync()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


for label, cuts in (("one launch", [0, nq]),
                    ("14 slices like ffx_score's (ramp 0.5 M .. 4 M pairs .. ramp down)", None),
                    ("8 equal slices", [nq * i // 8 for i in range(9)]),
                    ("32 equal slices", [nq * i // 32 for i in range(33)])):
    if cuts is None:
        sizes = [104, 209, 419] + [838] * 6 + [610, 305, 152, 104, 49]  # queries per slice of the traced call
        sizes[-1] += nq - sum(sizes)
        cuts = np.concatenate([[0], np.cumsum(sizes)]).tolist()
    plans = []
    stream = None
    for a, b in zip(cuts[:-1], cuts[1:]):
        p = native.Plan(index)
        if stream is None:
            stream = p.stream()
        else:
            p.set_stream(stream)  # every slice's kernels on the first plan's stream, in order
        p.set_queries(qv[a:b])
        off = (np.arange(b - a + 1, dtype=np.int64)) * k
        p.set_pairs(off, ids[a * k:b * k], sparse[a * k:b * k], 2)  # MaxP
        plans.append(p)
    ms = timed(plans)
    gather = sum(p.stats()["ms_gather_dot"] for p in plans)
    out[label] = {"slices": len(plans), "wall_ms": ms, "sum_gather_dot_ms": gather, "sum_device_ms": sum(p.stats()["ms_total"] for p in plans)}
    for p in plans:
        p.close()
print(json.dumps(out, indent=1))
