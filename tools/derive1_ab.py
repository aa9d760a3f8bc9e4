#!/usr/bin/env python3
"""A/B of the row-deriving headline kernel's tile (rows per wave) and rows in flight (experiments build: FFX_DEBUG_VARIANT=derive_tile=N,
derive_u=N), interleaved three times, HBM-resident step at the headline size."""
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
if len(sys.argv) > 1 and sys.argv[1] == "ragged":
    # the same candidates cut into queries of 1 .. 2 k candidates (a first stage that returns what it finds): the proportional
    # guess for a tile's query is off by hundreds of queries in the middle of the batch
    rng = np.random.default_rng(5)
    lens = rng.integers(1, 2 * k, size=nq)
    lens = (lens * (len(ids) / lens.sum())).astype(np.int64)
    lens[-1] += len(ids) - lens.sum()
    assert lens.min() >= 0
    pair_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
queries = synth.rows(2027, scale, 0, nq, d)
plan = native.Plan(idx)
plan.set_queries(queries)
plan.set_pairs(pair_off, ids, sparse, 2)
variants = {"shipping (tile 8, 4 in flight)": "", "row list (no_derive)": "no_derive", "tile 16": "derive_tile=16", "tile 32": "derive_tile=32", "tile 4": "derive_tile=4",
            "tile 8, 8 in flight": "derive_tile=8,derive_u=8", "tile 16, 8 in flight": "derive_tile=16,derive_u=8", "tile 8, 2 in flight": "derive_tile=8,derive_u=2"}
out = {k: [] for k in variants}
want = None
for rep in range(3):
    for label, v in variants.items():
        if v:
            os.environ["FFX_DEBUG_VARIANT"] = v
        else:
            os.environ.pop("FFX_DEBUG_VARIANT", None)
        plan.run(0.1, exact_sum=True, sort=True)
        plan.sync()
        ms = []
        for _ in range(6):
            plan.run(0.1, exact_sum=True, sort=True)
            plan.sync()
            ms.append(plan.stats()["ms_gather_dot"])
        ff = plan.fetch()[0]
        if want is None:
            want = ff.copy()
        assert np.array_equal(ff.view(np.uint32), want.view(np.uint32)), label
        out[label].append(round(float(np.median(ms)), 3))
print(json.dumps(out, indent=1))
