#!/usr/bin/env python3
"""One ffx_score call at the headline size under `rocprofv3 --kernel-trace --memory-copy-trace`: run this script under the
profiler, then `python tools/score_timeline.py --read <dir>` prints the device timeline of the LAST call (kernels per slice,
gaps on the kernel stream, head before the first kernel, tail after the last)."""
import csv
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "fast-forward-indexes_amd"))
sys.path.insert(0, str(ROOT))


def read(d):
    K = [r for f in Path(d).rglob("*kernel_trace.csv") for r in csv.DictReader(open(f))]
    M = [r for f in Path(d).rglob("*memory_copy_trace.csv") for r in csv.DictReader(open(f))]
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]) for r in K]
    ev.sort()
    g = [i for i, e in enumerate(ev) if "gather_dot" in e[2]]
    # calls are separated by long idle stretches: take the last run of gather_dot kernels with gaps < 3 ms
    last = [g[-1]]
    for i in reversed(g[:-1]):
        if ev[last[0]][0] - ev[i][1] < 3e6:
            last.insert(0, i)
        else:
            break
    t0, t1 = ev[last[0]][0], ev[last[-1]][1]
    window = [e for e in ev if e[0] >= t0 - 3e6 and e[0] <= t1 + 1e6]
    first = min(e[0] for e in window)
    copies = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"]) for r in M if first - 1e6 <= int(r["Start_Timestamp"]) <= t1 + 5e6]
    c0 = min(c[0] for c in copies)
    c1 = max(c[1] for c in copies)
    busy = sum(e[1] - e[0] for e in window)
    by = {}
    for e in window:
        by[e[2]] = by.get(e[2], 0) + (e[1] - e[0])
    gaps = sorted(((window[i + 1][0] - max(x[1] for x in window[: i + 1])) for i in range(len(window) - 1)), reverse=True)
    print(json.dumps({
        "slices": len(last), "first_copy_to_last_copy_ms": (c1 - c0) / 1e6, "first_copy_to_first_kernel_ms": (window[0][0] - c0) / 1e6,
        "first_kernel_to_last_kernel_end_ms": (max(e[1] for e in window) - window[0][0]) / 1e6, "last_kernel_to_last_copy_ms": (c1 - max(e[1] for e in window)) / 1e6,
        "kernel_busy_ms": busy / 1e6, "kernel_ms_by_name": {k: v / 1e6 for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:8]},
        "largest_gaps_us": [x / 1e3 for x in gaps[:12]], "sum_positive_gaps_ms": sum(x for x in gaps if x > 0) / 1e6,
        "gather_dot_start_end_ms_since_first_copy": [[round((ev[i][0] - c0) / 1e6, 3), round((ev[i][1] - c0) / 1e6, 3)] for i in last],
        "copies": len(copies), "copy_busy_ms_by_direction": {d: sum(c[1] - c[0] for c in copies if c[2] == d) / 1e6 for d in sorted({c[2] for c in copies})},
    }, indent=1))


if len(sys.argv) > 2 and sys.argv[1] == "--read":
    read(sys.argv[2])
    sys.exit(0)

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
if len(sys.argv) > 1 and sys.argv[1] == "ordinary":  # ordinary memory both ways (staged by the library: csrc/host_xfer.h, staged_d2h)
    p_q, p_ids, p_sp = queries, ids, sparse
    outs = (np.zeros(n, dtype=np.float32), np.zeros(n, dtype=np.float32), np.zeros(n, dtype=np.int32))
ts = []
for _ in range(4):
    time.sleep(0.02)
    t0 = time.perf_counter()
    native.score(idx, p_q, pair_off, p_ids, 2, p_sp, 0.1, exact_sum=True, sort=True, out=outs)
    ts.append((time.perf_counter() - t0) * 1e3)
print(json.dumps({"call_ms": ts}))
