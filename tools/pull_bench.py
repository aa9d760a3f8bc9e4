#!/usr/bin/env python3
"""Rate of ffx_index_pull_rows_async (the device copying scattered rows out of page-locked host memory) by where the
chunks start: 16-byte aligned (16-byte loads), 8- and 4-byte aligned, odd (single bytes) -- an HDF5 file places its chunks
wherever its allocator got to."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "fast-forward-indexes_amd"))
from fast_forward_amd import native  # noqa: E402

dim, chunk_rows, n_chunks, n = 768, 16384, 24, 170_000
rng = np.random.default_rng(0)
chunk_bytes = chunk_rows * dim * 4
out = {}
dev = native.DeviceIndex(n, dim)
for shift in (0, 8, 4, 2, 1):
    stride = chunk_bytes + 4096
    block = native.pinned_empty(n_chunks * stride + 64, np.uint8)
    block[:] = 1
    chunks = [block[c * stride + shift:c * stride + shift + chunk_bytes].view(np.float32).reshape(chunk_rows, dim) if shift % 4 == 0 else None for c in range(n_chunks)]
    if chunks[0] is None:  # float32 views need 4-byte alignment in NumPy: use a float16 / uint8 index for the odd starts
        continue
    rows = np.sort(rng.permutation(chunk_rows * n_chunks)[:n])  # distinct rows: a row fetched twice is served by the L2
    for _ in range(2):
        dev.pull_rows(chunks, rows, 0)
        dev.upload_wait()
    t0 = time.perf_counter()
    for _ in range(5):
        dev.pull_rows(chunks, rows, 0)
        dev.upload_wait()
    dt = (time.perf_counter() - t0) / 5
    out[f"chunks_start_at_16k+{shift}"] = {"GBs": n * dim * 4 / dt / 1e9, "ms": dt * 1e3}
    del chunks, block
dev.close()
# single-byte units: a uint8 index of 3072-byte rows at an odd start
dev = native.DeviceIndex(n, 3072, dtype=np.uint8)
for shift in (1, 2):
    stride = chunk_rows * 3072 + 4096
    block = native.pinned_empty(n_chunks * stride + 64, np.uint8)
    block[:] = 1
    chunks = [block[c * stride + shift:c * stride + shift + chunk_rows * 3072].reshape(chunk_rows, 3072) for c in range(n_chunks)]
    rows = np.sort(rng.permutation(chunk_rows * n_chunks)[:n])
    dev.pull_rows(chunks, rows, 0)
    dev.upload_wait()
    t0 = time.perf_counter()
    for _ in range(3):
        dev.pull_rows(chunks, rows, 0)
        dev.upload_wait()
    dt = (time.perf_counter() - t0) / 3
    out[f"bytes_rows_start_at_16k+{shift}"] = {"GBs": n * 3072 / dt / 1e9, "ms": dt * 1e3}
    del chunks, block
dev.close()
print(json.dumps(out, indent=1))
