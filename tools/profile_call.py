#!/usr/bin/env python3
"""cProfile of Index.__call__ / rerank on a pandas Ranking (region iii: DESIGN.md section 7)."""
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
from fast_forward_amd import Ranking, synth  # noqa: E402
from fast_forward_amd.encoder import LambdaEncoder  # noqa: E402
from fast_forward_amd.index import InMemoryIndex, Mode  # noqa: E402
import bench as B  # noqa: E402

n, nq, k, d = 1_000_000, 1000, 1000, 768
scale = float(1 / np.sqrt(d))
qv = synth.rows(5, scale, 0, nq, d)
index = InMemoryIndex(LambdaEncoder(lambda t: qv[int(t)]), mode=Mode.MAXP, init_size=n)
for lo in range(0, n, 100_000):
    index.add(synth.rows(1, scale, lo, 100_000, d), doc_ids=[f"D{i}" for i in range(lo, lo + 100_000)])
cid, csp = B.candidates(n, nq, k, 3)
df = pd.DataFrame({"q_id": np.repeat([f"q{i:05d}" for i in range(nq)], k), "id": [f"D{i}" for i in cid], "score": csp})
first = Ranking(df, queries={f"q{i:05d}": str(i) for i in range(nq)})
index(first)
for name, fn in (("call", lambda: index(first)), ("rerank", lambda: index.rerank(first, 0.1))):
    t0 = time.perf_counter()
    fn()
    print(name, "seconds", time.perf_counter() - t0)
    pr = cProfile.Profile()
    pr.enable()
    fn()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
