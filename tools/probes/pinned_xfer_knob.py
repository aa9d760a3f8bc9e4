#!/usr/bin/env python3
"""What GPU_PINNED_MIN_XFER_SIZE does in this runtime: blocking hipMemcpy between ordinary host memory and the device, sizes
0.5 MiB .. 256 MiB, GB/s both ways (the environment variable is read when the runtime starts: one process per value)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

hip = C.CDLL("libamdhip64.so")
dev = C.c_void_p()
assert hip.hipMalloc(C.byref(dev), C.c_size_t(256 << 20)) == 0
out = {"GPU_PINNED_MIN_XFER_SIZE": os.environ.get("GPU_PINNED_MIN_XFER_SIZE")}
for mb in (0.5, 1.5, 4, 16, 24, 48, 128, 256):
    n = int(mb * (1 << 20))
    a = np.ones(n, dtype=np.uint8)  # ordinary memory, touched
    best = [1e9, 1e9]
    for _ in range(4):
        for k, kind in enumerate((1, 2)):  # H2D, D2H
            t0 = time.perf_counter()
            rc = hip.hipMemcpy(dev, C.c_void_p(a.ctypes.data), C.c_size_t(n), kind) if kind == 1 else hip.hipMemcpy(C.c_void_p(a.ctypes.data), dev, C.c_size_t(n), kind)
            assert rc == 0
            best[k] = min(best[k], time.perf_counter() - t0)
    out[f"{mb} MiB"] = {"h2d_GBs": round(n / best[0] / 1e9, 1), "d2h_GBs": round(n / best[1] / 1e9, 1)}
print(json.dumps(out))
