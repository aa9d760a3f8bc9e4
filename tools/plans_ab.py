#!/usr/bin/env python3
"""ffx_score at the headline size: how many plans should the slices rotate through?  Page-locked and ordinary buffers, 3 / 4 / 5 plans
(FFX_DEBUG_VARIANT=score_plans=N: the experiments build), best and median of 9 calls each, interleaved twice."""
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
p_out = tuple(native.pinned_empty(n, t) for t in (np.float32, np.float32, np.int32))
o_out = (np.zeros(n, dtype=np.float32), np.zeros(n, dtype=np.float32), np.zeros(n, dtype=np.int32))
forms = {"page_locked": lambda: native.score(idx, p_q, pair_off, p_ids, 2, p_sp, 0.1, exact_sum=True, sort=True, out=p_out),
         "ordinary": lambda: native.score(idx, queries, pair_off, ids, 2, sparse, 0.1, exact_sum=True, sort=True, out=o_out)}
out = {}
for rep in range(2):
    for name, run in forms.items():
        for plans in (3, 4, 5):
            os.environ["FFX_DEBUG_VARIANT"] = f"score_plans={plans}"
            run()
            run()
            ts = []
            for _ in range(9):
                t0 = time.perf_counter()
                run()
                ts.append(time.perf_counter() - t0)
            out[f"{name} plans={plans} pass {rep}"] = {"best_ms": round(min(ts) * 1e3, 3), "median_ms": round(float(np.median(ts)) * 1e3, 3)}
print(json.dumps(out, indent=1))
