#!/usr/bin/env python3
"""Where the FIRST ffx_score_ranked call of a process spends its time (region (iii) cold: Index.rerank's device part), at the
headline size: FFX_SCORE_TRACE=1 timeline of the cold call, then the warm one.  `pool` / `pageable` as argv[1]: results into
page-locked blocks of the result pool (reserved before the call, as Index.rerank does while it looks the ids up) or into fresh
ordinary arrays."""
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

how = sys.argv[1] if len(sys.argv) > 1 else "pool"
rows, d, nq, k = 8_841_823, 768, 6980, 5000
scale = float(1 / np.sqrt(d))
idx = native.DeviceIndex(rows, d)
idx.fill_synthetic(2024, scale, 0, rows)
idx.set_doc_rows(np.arange(rows + 1, dtype=np.int64), np.arange(rows, dtype=np.int32))
idx.set_psg_rows(np.arange(rows, dtype=np.int32))
ids, sparse = B.candidates(rows, nq, k, 2024)
pair_off = np.arange(nq + 1, dtype=np.int64) * k
queries = synth.rows(2027, scale, 0, nq, d)
carry = np.arange(nq * k, dtype=np.int32)
out = {"results": how}
t0 = time.perf_counter()
if how == "pool":
    native.reserve_pooled([(nq * k, np.float32), (nq * k, np.int32)]).join()
out["reserve_s"] = time.perf_counter() - t0
os.environ["FFX_SCORE_TRACE"] = "1"
for label in ("cold", "warm", "warm2"):
    print(f"---- {label}", file=sys.stderr)
    t0 = time.perf_counter()
    res = native.score_ranked(idx, queries, pair_off, ids, 2, sparse, 0.1, carry=carry, pool=(how == "pool"))
    out[f"{label}_s"] = time.perf_counter() - t0
    if label != "warm":
        del res
print(json.dumps(out))
