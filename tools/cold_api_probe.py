#!/usr/bin/env python3
"""Region (iii) COLD, taken apart: Ranking(frame) of 34.9 M string-id rows, then the first Index.rerank on it under cProfile
with FFX_SCORE_TRACE=1 (the slices' host-side timeline on stderr), then two warm calls and index(first) + interpolate."""
import cProfile
import json
import os
import pstats
import sys
import time
from pathlib import Path

import numpy as np
import pandas as pd

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "fast-forward-indexes_amd"))
sys.path.insert(0, str(ROOT))
from fast_forward_amd import Ranking, native, synth  # noqa: E402
from fast_forward_amd.encoder import ArrayEncoder  # noqa: E402
from fast_forward_amd.index import InMemoryIndex, Mode  # noqa: E402
import bench as B  # noqa: E402

rows, d, nq, k = 8_841_823, 768, 6980, 5000
scale = float(1 / np.sqrt(d))
store = native.DeviceIndex(rows, d)
store.fill_synthetic(2024, scale, 0, rows)
doc_names = np.array([f"D{i}" for i in range(rows)], dtype=object)
api = InMemoryIndex.adopt(store, doc_ids=doc_names, doc_off=np.arange(rows + 1, dtype=np.int64), mode=Mode.MAXP)
ids, sparse = B.candidates(rows, nq, k, 2024)
qvecs = synth.rows(2027, scale, 0, nq, d)
q_names = np.array([f"q{i:06d}" for i in range(nq)], dtype=object)
api.query_encoder = ArrayEncoder(dict(zip(q_names.tolist(), qvecs)))
frame = pd.DataFrame({"q_id": np.repeat(q_names, k), "id": doc_names[ids], "score": sparse, "query": np.repeat(q_names, k)})
out = {}
t0 = time.perf_counter()
first = Ranking(frame, name="first-stage", copy=False)
out["ranking_constructor_s"] = time.perf_counter() - t0
os.environ["FFX_SCORE_TRACE"] = "1"
print("---- cold rerank", file=sys.stderr)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
final = api.rerank(first, 0.1)
pr.disable()
out["rerank_first_call_s"] = time.perf_counter() - t0
out["first_call_breakdown"] = dict(api.last_timings)
pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(30)
os.environ.pop("FFX_SCORE_TRACE")
for label in ("rerank_second_s", "rerank_third_s"):
    t0 = time.perf_counter()
    final = api.rerank(first, 0.1)
    out[label] = time.perf_counter() - t0
t0 = time.perf_counter()
dense = api(first)
t1 = time.perf_counter()
both = first.interpolate(dense, 0.1)
out["call_s"], out["interpolate_s"] = t1 - t0, time.perf_counter() - t1
out["interpolate_breakdown"] = dict(api.last_timings)
t1 = time.perf_counter()
both = first.interpolate(dense, 0.3)
out["interpolate_again_s"] = time.perf_counter() - t1
print(json.dumps(out))
