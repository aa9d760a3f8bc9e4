#!/usr/bin/env python3
"""Row reuse probe.  At k = 5000 every index row is wanted by ~4 of the 6980 queries.  gather_dot keeps the
QUERY in registers and gathers rows (one HBM fetch per pair).  With the roles swapped -- the 8.8 M document
rows are the "queries" (kept in registers, one HBM fetch per DOCUMENT), the 6980 query vectors (21 MB) are the
"index" and get gathered (from the Infinity Cache / L2) -- the same 34.9 M dot products need a quarter of
the HBM bytes.  Same kernel, same bits (a product does not care which factor is called the query)."""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "fast-forward-indexes_amd"))
sys.path.insert(0, str(ROOT))
from fast_forward_amd import native, synth  # noqa: E402
import bench as B  # noqa: E402

n, d, nq, k = int(os.environ.get("PROBE_ROWS", 8_841_823)), 768, 6980, 5000
docs = native.DeviceIndex(n, d)
docs.fill_synthetic(2024, 0.036, 0, n)
docs.set_doc_rows(np.arange(n + 1, dtype=np.int64), np.arange(n, dtype=np.int32))
docs.set_psg_rows(np.arange(n, dtype=np.int32))
ids, sparse = B.candidates(n, nq, k, 2024)
pair_off = np.arange(nq + 1, dtype=np.int64) * k
qv = synth.rows(2027, 0.036, 0, nq, d)
out = {}
# (a) the shipping orientation
plan = native.Plan(docs)
plan.set_queries(qv)
plan.set_pairs(pair_off, ids, None, 1)
ms = []
for _ in range(6):
    plan.run(0.0, exact_sum=True)
    ms.append(plan.stats()["ms_gather_dot"])
ff = plan.fetch()[0]
out["query_in_registers_rows_gathered"] = {"ms": float(np.mean(ms[1:]))}
plan.close()
# (b) transposed: documents in registers, query vectors gathered
t0 = time.perf_counter()
by_doc = np.argsort(ids, kind="stable")
t_pairs = ids[by_doc]                                   # "query number" = document, ascending
t_ids = (by_doc // k).astype(np.int32)                  # "row" = original query number
t_off = np.zeros(n + 1, dtype=np.int64)
np.cumsum(np.bincount(t_pairs, minlength=n), out=t_off[1:])
out["host_transpose_s"] = time.perf_counter() - t0
qidx = native.DeviceIndex(nq, d)
qidx.upload(qv, 0)
qidx.set_doc_rows(np.arange(nq + 1, dtype=np.int64), np.arange(nq, dtype=np.int32))
qidx.set_psg_rows(np.arange(nq, dtype=np.int32))
tplan = native.Plan(qidx)
tplan.set_queries_device(docs.device_ptr(), n)
tplan.set_pairs(t_off, t_ids, None, 1)
for variant in ("", "914", "913", "911", "900"):
    if variant:
        os.environ["FFX_DEBUG_VARIANT"] = f"gather={variant}"
    else:
        os.environ.pop("FFX_DEBUG_VARIANT", None)
    ms = []
    for _ in range(6):
        tplan.run(0.0, exact_sum=True)
        ms.append(tplan.stats()["ms_gather_dot"])
    got = tplan.fetch()[0]
    same = bool(np.array_equal(got.view(np.uint32), ff[by_doc].view(np.uint32)))
    out[f"docs_in_registers_queries_gathered_variant_{variant or 'default'}"] = {"ms": float(np.mean(ms[1:])), "bit_identical": same,
                                                                               "kernel_variant": tplan.stats()["kernel_variant"]}
os.environ.pop("FFX_DEBUG_VARIANT", None)
print(json.dumps(out, indent=1))
