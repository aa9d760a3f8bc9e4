#!/usr/bin/env python3
"""ffx_score at the headline size with ORDINARY (not page-locked) buffers both ways, written before: best of 5 calls and the
host-side timeline of one (FFX_SCORE_TRACE=1).  argv[1]: 'in' = only the inputs ordinary, 'out' = only the outputs, 'both'; argv[2:]: the
FFX_DEBUG_VARIANT settings to time one after the other (experiments build)."""
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

which = sys.argv[1] if len(sys.argv) > 1 else "both"
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
if which == "out":
    pq = native.pinned_empty(queries.shape, np.float32)
    pi, ps = native.pinned_empty(n, np.int32), native.pinned_empty(n, np.float32)
    pq[...], pi[...], ps[...] = queries, ids, sparse
    queries, ids, sparse = pq, pi, ps
outs = tuple(native.pinned_empty(n, t) for t in (np.float32, np.float32, np.int32)) if which == "in" else \
    (np.zeros(n, dtype=np.float32), np.zeros(n, dtype=np.float32), np.zeros(n, dtype=np.int32))
run = lambda: native.score(idx, queries, pair_off, ids, 2, sparse, 0.1, exact_sum=True, sort=True, out=outs)  # noqa: E731
run()
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    run()
    ts.append(time.perf_counter() - t0)
print(json.dumps({"ordinary": which, "best_ms": min(ts) * 1e3, "median_ms": float(np.median(ts)) * 1e3}))
VARIANTS = sys.argv[2:] or ["score_plans=3", "score_plans=4", "score_plans=5", "score_ramp_growth_pct=150", "score_ramp_growth_pct=150,score_ramp_pairs=400000",
                            "score_ramp_growth_pct=170", "score_ramp_growth_pct=130,score_ramp_pairs=400000"]
for plans in VARIANTS:  # (argv[2:]: the switches to try instead, e.g. xfer_lanes=2 xfer_lanes=4; "none" = no switch; "trace:<switches>" = one traced call)
    traced = plans.startswith("trace:")
    setting = plans[6:] if traced else plans
    if setting in ("", "none"):
        os.environ.pop("FFX_DEBUG_VARIANT", None)
    else:
        os.environ["FFX_DEBUG_VARIANT"] = setting
    run()
    if traced:
        print(f"--- timeline with {setting or 'none'}", flush=True)
        os.environ["FFX_SCORE_TRACE"] = "1"
        run()
        os.environ.pop("FFX_SCORE_TRACE")
        continue
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        run()
        ts.append(time.perf_counter() - t0)
    print(json.dumps({"variant": plans, "best_ms": min(ts) * 1e3, "median_ms": float(np.median(ts)) * 1e3}), flush=True)
os.environ.pop("FFX_DEBUG_VARIANT", None)
if len(sys.argv) <= 2:
    os.environ["FFX_SCORE_TRACE"] = "1"
    run()
