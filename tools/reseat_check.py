#!/usr/bin/env python3
"""ffx_index_reseat under the lamp: the headline index is scored, re-seated, probed again where it ended up and scored again -- does the
block keep, with the rows in it, the figure it had as an empty candidate?  Prints one JSON line per trial."""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "fast-forward-indexes_amd"))
sys.path.insert(0, str(ROOT))
from fast_forward_amd import native, synth  # noqa: E402
import bench as B  # noqa: E402

tries = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rows, d, nq, k = 8_841_823, 768, 6980, 5000
scale = float(1 / np.sqrt(d))
ids, sparse = B.candidates(rows, nq, k, 2024)
pair_off = np.arange(nq + 1, dtype=np.int64) * k
queries = synth.rows(2027, scale, 0, nq, d)


def kernel_ms(plan):
    ms = []
    for _ in range(4):
        plan.run(0.1, exact_sum=True, sort=True)
        plan.sync()
        ms.append(plan.stats()["ms_gather_dot"])
    return round(min(ms[1:]), 3)


for trial in range(2):
    idx = native.DeviceIndex(rows, d)
    idx.fill_synthetic(2024, scale, 0, rows)
    idx.set_doc_rows(np.arange(rows + 1, dtype=np.int64), np.arange(rows, dtype=np.int32))
    idx.set_psg_rows(np.arange(rows, dtype=np.int32))
    plan = native.Plan(idx)
    plan.set_queries(queries)
    plan.set_pairs(pair_off, ids, sparse, 2)
    before = kernel_ms(plan)
    probe_before = round(idx.probe_placement(), 4)
    rep = idx.reseat(tries)
    probe_after = round(idx.probe_placement(), 4)
    after = kernel_ms(plan)
    # ... and as time goes by (is what the freed blocks leave behind a passing thing?)
    import time
    later = {}
    t0 = time.perf_counter()
    for wait in (0.5, 1, 2, 4, 8):
        time.sleep(max(0.0, wait - (time.perf_counter() - t0)))
        later[f"+{wait}s"] = kernel_ms(plan)
    print(json.dumps({"trial": trial, "kernel_ms_before": before, "probe_ms_before": probe_before, "reseat": rep, "probe_ms_where_it_ended_up": probe_after,
                      "kernel_ms_after": after, "kernel_ms_later": later}), flush=True)
    plan.close()
    idx.close()
