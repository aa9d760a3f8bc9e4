#!/usr/bin/env python3
"""Is a slow allocation slow everywhere?  K copies of the headline index alive in one process; the whole step on each, then the same
kernel over candidates drawn from one EIGHTH of the rows at a time (2 M candidates: ~0.9 ms).  Prints one JSON line."""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "fast-forward-indexes_amd"))
sys.path.insert(0, str(ROOT))
from fast_forward_amd import native, synth  # noqa: E402
import bench as B  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
PARTS = 8
rows, d, nq, k = 8_841_823, 768, 6980, 5000
scale = float(1 / np.sqrt(d))
ids, sparse = B.candidates(rows, nq, k, 2024)
pair_off = np.arange(nq + 1, dtype=np.int64) * k
queries = synth.rows(2027, scale, 0, nq, d)
rng = np.random.default_rng(11)
sub_q = 400
sub_off = np.arange(sub_q + 1, dtype=np.int64) * k
sub_ids = [rng.integers(p * rows // PARTS, (p + 1) * rows // PARTS, size=sub_q * k).astype(np.int32) for p in range(PARTS)]
sub_sparse = np.zeros(sub_q * k, dtype=np.float32)


def best_ms(plan, sort):
    ms = []
    for _ in range(4):
        plan.run(0.1, exact_sum=True, sort=sort)
        plan.sync()
        ms.append(plan.stats()["ms_gather_dot"])
    return min(ms[1:])


out = []
keep = []
for c in range(K):
    idx = native.DeviceIndex(rows, d)
    idx.fill_synthetic(2024, scale, 0, rows)
    idx.set_doc_rows(np.arange(rows + 1, dtype=np.int64), np.arange(rows, dtype=np.int32))
    idx.set_psg_rows(np.arange(rows, dtype=np.int32))
    keep.append(idx)
for c, idx in enumerate(keep):
    plan = native.Plan(idx)
    plan.set_queries(queries)
    plan.set_pairs(pair_off, ids, sparse, 2)
    full = best_ms(plan, True)
    plan.close()
    parts = []
    sub = native.Plan(idx)
    sub.set_queries(queries[:sub_q])
    for p in range(PARTS):
        sub.set_pairs(sub_off, sub_ids[p], sub_sparse, 1)  # PASSAGE: ordinals are rows (duplicates inside a query do not matter here)
        parts.append(round(best_ms(sub, False), 4))
    sub.close()
    out.append({"copy": c, "full_ms": round(full, 3), "eighths_ms": parts, "device_ptr": hex(idx.device_ptr())})
print(json.dumps({"copies": out}))
