#!/usr/bin/env python3
"""Large pageable transfers: plain runtime copies (FFX_NO_BIG_COPY=1) vs the threaded staged copy."""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "fast-forward-indexes_amd"))
from fast_forward_amd import native  # noqa: E402

n, d, nq, k = 1_000_000, 64, 6980, 5000  # 34.9 M pairs: 140 MB per array
idx = native.DeviceIndex(n, d)
idx.fill_synthetic(1, 0.1, 0, n)
idx.set_doc_rows(np.arange(n + 1), np.arange(n))
idx.set_psg_rows(np.arange(n))
rng = np.random.default_rng(0)
pair_off = np.arange(nq + 1, dtype=np.int64) * k
ids = rng.integers(0, n, size=nq * k).astype(np.int32)
sparse = rng.random(nq * k).astype(np.float32)
plan = native.Plan(idx)
plan.set_queries(rng.standard_normal((nq, d)).astype(np.float32))
out = {}
for label, env in (("plain", "1"), ("staged", None), ("plain_again", "1"), ("staged_again", None)):
    if env:
        os.environ["FFX_NO_BIG_COPY"] = env
    else:
        os.environ.pop("FFX_NO_BIG_COPY", None)
    t0 = time.perf_counter()
    plan.set_pairs(pair_off, ids, sparse, 1)
    t1 = time.perf_counter()
    plan.run(0.1, sort=True)
    plan.sync()
    t2 = time.perf_counter()
    plan.fetch(interp=True, order=True)
    t3 = time.perf_counter()
    out[label] = {"set_pairs_ms (280 MB up + validation)": (t1 - t0) * 1e3, "run_ms": (t2 - t1) * 1e3, "fetch_ms (420 MB down)": (t3 - t2) * 1e3}
print(json.dumps(out, indent=1))
