#!/usr/bin/env python3
"""Known-byte-count launches for calibrating rocprofv3's FETCH_SIZE on gfx950 in THIS kernel's
access pattern (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reads 1/2 of a wide coalesced stream;
other widths must be calibrated).  Run under `rocprofv3 --pmc ...`; dispatch order is fixed:

  1. gather_dot fast  (dwordx4),  rows 0..N-1 in order, each exactly once   -> N*d*4 bytes, streaming
  2. gather_dot exact (dword),    rows 0..N-1 in order, each exactly once   -> N*d*4 bytes, streaming
  3. gather_dot fast,  N random distinct rows (a permutation)               -> N*d*4 bytes, gathered
  4. gather_dot exact, N random distinct rows (a permutation)               -> N*d*4 bytes, gathered

Every row is touched once per launch and the index (N*d*4 = 6.3 GB by default) is far larger than
the 256 MiB Infinity Cache, so each launch must move exactly N*d*4 bytes of row data from HBM.
"""
import argparse
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "fast-forward-indexes_amd"))
from fast_forward_amd import native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=2_048_000)
ap.add_argument("--dim", type=int, default=768)
args = ap.parse_args()

n, d = args.rows, args.dim
idx = native.DeviceIndex(n, d)
idx.fill_synthetic(1, 0.03, 0, n)
idx.set_doc_rows(np.arange(n + 1), np.arange(n))
idx.set_psg_rows(np.arange(n))
queries = np.ones((4, d), dtype=np.float32)
pair_off = np.array([0, n // 4, n // 2, 3 * (n // 4), n], dtype=np.int64)
perm = np.random.default_rng(0).permutation(n).astype(np.int32)
plan = native.Plan(idx)
plan.set_queries(queries)
out = []
for ids, label in ((np.arange(n, dtype=np.int32), "sequential"), (perm, "permuted")):
    plan.set_pairs(pair_off, ids, None, 1)
    for exact in (False, True):
        plan.run(0.0, exact_sum=exact)
        st = plan.stats()
        out.append({"order": label, "sum": "exact" if exact else "fast", "variant": st["kernel_variant"],
                    "row_bytes": n * d * 4, "ms": st["ms_gather_dot"], "GBs": n * d * 4 / st["ms_gather_dot"] / 1e6})
print(json.dumps(out))
