#!/usr/bin/env python3
"""Per-query ordering: rocPRIM's segmented radix sort vs the in-LDS block sort, per candidate-list length
(6980 queries each).  FFX_DEBUG_VARIANT=sort_block_from=N / sort_block_to=N select which one a plan uses."""
import json
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "fast-forward-indexes_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import _experiments  # noqa: E402

_experiments.use()  # the switches this tool flips exist in the experiments build only
from fast_forward_amd import native  # noqa: E402

n, d = 200_000, 64
PAIRS = 7_000_000  # per case: queries = PAIRS / k
idx = native.DeviceIndex(n, d)
idx.fill_synthetic(1, 0.1, 0, n)
idx.set_doc_rows(np.arange(n + 1), np.arange(n))
idx.set_psg_rows(np.arange(n))
rng = np.random.default_rng(0)
out = {}
for k in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "10,50,100,256,500,1000,2000,5000,8192").split(",")]:
    nq = max(1, PAIRS // k)
    queries = rng.standard_normal((nq, d)).astype(np.float32)
    pair_off = np.arange(nq + 1, dtype=np.int64) * k
    ids = rng.integers(0, n, size=nq * k).astype(np.int32)
    sparse = (rng.random(nq * k) * 25).astype(np.float32)
    plan = native.Plan(idx)
    plan.set_queries(queries)
    plan.set_pairs(pair_off, ids, sparse, 1)
    res = {}
    for label, env in (("rocprim_segmented", {"FFX_DEBUG_VARIANT": "sort_block_from=1000000"}), ("block_lds", {"FFX_DEBUG_VARIANT": "sort_block_from=1"})):
        os.environ.update(env)
        ms = []
        for _ in range(6):
            plan.run(0.1, sort=True)
            ms.append(plan.stats()["ms_sort"])
        res[label] = {"ms": float(np.mean(ms[1:])), "order_checksum": int(plan.fetch(order=True)[2].astype(np.int64)[::7].sum())}
    res["same_order"] = res["rocprim_segmented"]["order_checksum"] == res["block_lds"]["order_checksum"]
    out[str(k)] = res
    plan.close()
print(json.dumps(out, indent=1))
