#!/usr/bin/env python3
"""gather_dot's time on the 27 GB index varies by ~3 % between PROCESSES (16.2 vs 16.7 ms, alternating from one launch to the
next: tools/locality_probe.py run eight times in a row), whatever the candidate order.  Does it vary between ALLOCATIONS inside
one process?  Allocate + fill + score + free the index several times and print gather_dot's time for each allocation, plus the
device address the runtime handed out."""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "fast-forward-indexes_amd"))
sys.path.insert(0, str(ROOT))
from fast_forward_amd import native, synth  # noqa: E402
import bench as B  # noqa: E402

n, d, nq, k = 8_841_823, 768, 6980, 5000
ids, sparse = B.candidates(n, nq, k, 2024)
off = np.arange(nq + 1, dtype=np.int64) * k
q = synth.rows(2027, 0.036, 0, nq, d)
out = []
hold = []
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    if len(sys.argv) > 2 and sys.argv[2] == "pad" and trial % 2 == 1:
        hold.append(native.DeviceIndex(1_000_000, d))  # something else takes the head of the free memory first
    idx = native.DeviceIndex(n, d)
    idx.fill_synthetic(2024, 0.036, 0, n)
    idx.set_doc_rows(np.arange(n + 1, dtype=np.int64), np.arange(n, dtype=np.int32))
    idx.set_psg_rows(np.arange(n, dtype=np.int32))
    plan = native.Plan(idx)
    plan.set_queries(q)
    plan.set_pairs(off, ids, sparse, 1)
    ms = []
    for _ in range(6):
        plan.run(0.1, exact_sum=True)
        ms.append(plan.stats()["ms_gather_dot"])
    out.append({"trial": trial, "device_ptr": hex(idx.device_ptr()), "gather_dot_ms": round(float(np.mean(ms[2:])), 3)})
    print(json.dumps(out[-1]), flush=True)
    plan.close()
    idx.close()
