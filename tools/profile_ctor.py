#!/usr/bin/env python3
"""cProfile of `Ranking(frame)` on a frame of string ids at headline scale (6980 queries x k = 5000 = 34.9 M rows, 8.67 M
distinct document names, objects shared between rows as a reader would leave them).  Host only: no GPU is touched."""
import cProfile
import pstats
import sys
import time
from pathlib import Path

import numpy as np
import pandas as pd

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "fast-forward-indexes_amd"))
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from fast_forward_amd import Ranking  # noqa: E402

nq, k, n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 6980, 5000, 8841823
t0 = time.perf_counter()
doc_names = np.array([f"D{i}" for i in range(n_docs)], dtype=object)
q_names = np.array([f"q{i:06d}" for i in range(nq)], dtype=object)
ids, sparse = bench.candidates(n_docs, nq, k, 2024)
frame = pd.DataFrame({"q_id": np.repeat(q_names, k), "id": doc_names[ids], "score": sparse, "query": np.repeat(q_names, k)})
print("frame built in", round(time.perf_counter() - t0, 2), "s")
pr = cProfile.Profile()
pr.enable()
t0 = time.perf_counter()
first = Ranking(frame, name="first-stage", copy=False)
print("Ranking(frame):", round(time.perf_counter() - t0, 3), "s,", "coded" if first._coded is not None else "frame")
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
