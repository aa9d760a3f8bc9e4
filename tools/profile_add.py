#!/usr/bin/env python3
"""cProfile of InMemoryIndex.add (the setup side of the path)."""
import cProfile
import pstats
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "fast-forward-indexes_amd"))
from fast_forward_amd import synth  # noqa: E402
from fast_forward_amd.index import InMemoryIndex, Mode  # noqa: E402

n, d, step = 1_000_000, 768, 100_000
index = InMemoryIndex(mode=Mode.MAXP, init_size=n)
batches = [(synth.rows(1, 0.03, lo, step, d), [f"D{i}" for i in range(lo, lo + step)], [f"P{i}" for i in range(lo, lo + step)]) for lo in range(0, n, step)]
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for vec, docs, psgs in batches:
    index.add(vec, doc_ids=docs, psg_ids=psgs)
pr.disable()
print("add seconds", time.perf_counter() - t0)
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
