#!/usr/bin/env python3
"""Does the ORDER of a query's candidates matter to gather_dot -- and is address translation what a 261 GB index pays for?
One configuration per process (so that a `rocprofv3 --pmc` pass of it counts that configuration only):

    locality_probe.py --rows N --order first|sorted|in_order [--steps K]

first:    candidate lists in first-stage order (random rows)
sorted:   every query's candidates sorted by row -- i.e. bucketed by 2 MB page, by HBM stack, by anything monotone in the address
          (a query's k = 5000 candidates over N >= 8.8 M rows are ~N / k rows apart: further than a page, whatever the order)
in_order: the whole batch's rows ascending through 4 pseudo-queries (every row once: the in-order ceiling of the box)
Prints gather_dot's time (HIP events) and algorithmic TB/s."""
import argparse
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "fast-forward-indexes_amd"))
sys.path.insert(0, str(ROOT))
from fast_forward_amd import native, synth  # noqa: E402
import bench as B  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=8_841_823)
ap.add_argument("--order", default="first", help="first | sorted | in_order | bucket:SHIFT (stable sort by row >> SHIFT inside every query)")
ap.add_argument("--steps", type=int, default=8)
args = ap.parse_args()
n, d, nq, k = args.rows, 768, 6980, 5000
idx = native.DeviceIndex(n, d)
idx.fill_synthetic(2024, 0.036, 0, n)
idx.set_doc_rows(np.arange(n + 1, dtype=np.int64), np.arange(n, dtype=np.int32))
idx.set_psg_rows(np.arange(n, dtype=np.int32))
plan = native.Plan(idx)
if args.order == "in_order":
    m = min(n, nq * k)
    plan.set_queries(synth.rows(2027, 0.036, 0, 4, d))
    off = np.array([0, m // 4, m // 2, 3 * (m // 4), m], dtype=np.int64)
    cand = (np.arange(m, dtype=np.int64) * (n // m)).astype(np.int32)
    plan.set_pairs(off, cand, None, 1)
    pairs = m
else:
    ids, sparse = B.candidates(n, nq, k, 2024)
    if args.order == "sorted":
        ids = np.sort(ids.reshape(nq, k), axis=1).reshape(-1)
    elif args.order.startswith("bucket:"):
        shift = int(args.order.split(":")[1])
        m = ids.reshape(nq, k)
        o = np.argsort(m >> shift, axis=1, kind="stable")
        ids = np.take_along_axis(m, o, axis=1).reshape(-1).copy()
    plan.set_queries(synth.rows(2027, 0.036, 0, nq, d))
    plan.set_pairs(np.arange(nq + 1, dtype=np.int64) * k, ids, sparse, 1)
    pairs = nq * k
ms = []
for _ in range(args.steps):
    plan.run(0.1, exact_sum=True)
    ms.append(plan.stats()["ms_gather_dot"])
use = ms[2:] if len(ms) > 3 else ms
print(json.dumps({"rows": n, "index_GB": n * d * 4 / 1e9, "order": args.order, "pairs": pairs, "gather_dot_ms": float(np.mean(use)),
                  "algorithmic_TBs": pairs * (d * 4 + 4) / float(np.mean(use)) / 1e9}))
