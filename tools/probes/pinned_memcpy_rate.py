#!/usr/bin/env python3
"""How fast the CPU copies OUT of page-locked (hipHostMalloc) memory into ordinary memory -- one thread, and several at once --
against an ordinary-to-ordinary copy: the second half of every staged device-to-host transfer."""
import json
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "fast-forward-indexes_amd"))
from fast_forward_amd import native  # noqa: E402

n = 64 << 20
out = {}
for label, make in (("page_locked_src", lambda: native.pinned_empty(n, np.uint8)), ("ordinary_src", lambda: np.empty(n, dtype=np.uint8))):
    for threads in (1, 3, 6):
        srcs = [make() for _ in range(threads)]
        dsts = [np.zeros(n, dtype=np.uint8) for _ in range(threads)]
        for s in srcs:
            s[...] = 7
        best = 1e9
        for _ in range(4):
            ts = [threading.Thread(target=np.copyto, args=(d, s)) for d, s in zip(dsts, srcs)]
            t0 = time.perf_counter()
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            best = min(best, time.perf_counter() - t0)
        out[f"{label} x{threads}"] = round(threads * n / best / 1e9, 1)
print(json.dumps(out))
