#!/usr/bin/env python3
"""Host-side timeline of one ffx_score call at the headline size (region ii, page-locked buffers):
FFX_SCORE_TRACE=1 makes the library print, per slice, how long the caller spent in set_queries /
set_pairs / run / fetch of the previous slice.  Also times the call for a few slice sizes."""
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

_experiments.use()  # the switches this tool flips exist in the experiments build only
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
outs = (native.pinned_empty(n, np.float32), native.pinned_empty(n, np.float32), native.pinned_empty(n, np.int32))
run = lambda: native.score(idx, p_q, pair_off, p_ids, 2, p_sp, 0.1, exact_sum=True, sort=True, out=outs)  # noqa: E731
run(); run()
out = {}


def best(n=4):
    run()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); run(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


def variant(*words):
    """FFX_DEBUG_VARIANT: libffx's one knob for A/B switches (csrc/ffx.hip, ffx_debug_variant)."""
    words = [w for w in words if w]
    if words:
        os.environ["FFX_DEBUG_VARIANT"] = ",".join(words)
    else:
        os.environ.pop("FFX_DEBUG_VARIANT", None)


for env_label, word in (("ramp", None), ("no_ramp", "score_no_ramp"), ("no_out_stream", "score_no_out_stream")):
    for label, pairs in (("default", None), ("1M", 1 << 20), ("4M", 4 << 20), ("8M", 8 << 20)):
        variant(word, None if pairs is None else f"score_slice_pairs={pairs}")
        out[f"{env_label}/{label}"] = best()
variant()
print(json.dumps(out))
# ordinary (pageable) arrays, written before
mine = (np.zeros(n, dtype=np.float32), np.zeros(n, dtype=np.float32), np.zeros(n, dtype=np.int32))
run_pinned = run
run = lambda: native.score(idx, queries, pair_off, ids, 2, sparse, 0.1, exact_sum=True, sort=True, out=mine)  # noqa: E731
pg = {}
for env_label, word in (("threaded", None), ("blocking", "score_blocking_out")):
    for label, pairs in (("default", None), ("2M", 2 << 20), ("4M", 4 << 20), ("8M", 8 << 20)):
        variant(word, None if pairs is None else f"score_slice_pairs={pairs}")
        pg[f"pageable/{env_label}/{label}"] = best(3)
variant()
print(json.dumps(pg))
same = all(np.array_equal(a, b) for a, b in zip(mine, outs))
print(json.dumps({"pageable_equals_pinned": bool(same)}))
run = run_pinned
print(json.dumps({"one_shot_stats_default_slices": run()[3]}))
variant(f"score_slice_pairs={4 << 20}")
run()
print(json.dumps({"one_shot_stats_4M_slices": run()[3]}))
variant()
run()
os.environ["FFX_SCORE_TRACE"] = "1"
run()
