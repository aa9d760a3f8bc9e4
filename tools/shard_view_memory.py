#!/usr/bin/env python3
"""Host memory of the shard views a ShardedOnDiskIndex opens (`OnDiskIndex.load(file, shard=(rank, world))`): a rank
reads the file's id columns block by block and interns only the ids of the rows it holds, so its private memory
(RssAnon: id tables + row map) goes with N / world, where loading the whole file costs O(N) on every rank.
CPU only (no GPU is touched): one fresh process per world size.

    python tools/shard_view_memory.py /dev/shm/probe.ffx [rows]
"""
import json
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "fast-forward-indexes_amd"))


def rss_anon_mb() -> float:
    for line in open("/proc/self/status"):
        if line.startswith("RssAnon:"):
            return int(line.split()[1]) / 1024.0
    return 0.0


if len(sys.argv) > 3 and sys.argv[2] == "--world":
    from fast_forward_amd import native
    from fast_forward_amd.index import OnDiskIndex

    native.lib()
    world = int(sys.argv[3])
    before = rss_anon_mb()
    view = OnDiskIndex.load(sys.argv[1], shard=(0, world) if world > 0 else None)
    print(json.dumps({"world": world or "whole file (no shard view)", "rows_held": len(view), "id_strings_interned": len(view.doc_ids) + len(view.psg_ids),
                      "private_memory_mb": round(rss_anon_mb() - before, 1)}))
else:
    from fast_forward_amd.index import Mode, OnDiskIndex

    path = Path(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
    w = OnDiskIndex(path, mode=Mode.MAXP, chunk_size=2**16, max_id_length=12, overwrite=True)
    for lo in range(0, n, 2**18):
        hi = min(n, lo + 2**18)
        w.add(np.zeros((hi - lo, 4), dtype=np.float32), doc_ids=[f"D{i // 3}" for i in range(lo, hi)], psg_ids=[f"P{i}" for i in range(lo, hi)])
    del w
    rows = [json.loads(subprocess.run([sys.executable, __file__, str(path), "--world", str(world)], capture_output=True, text=True, check=True).stdout)
            for world in (0, 1, 2, 4, 8)]
    path.unlink()
    print(json.dumps({"what": f"rank 0's view of a file of {n} rows (3 passages per document, a document id and a passage id per row)", "views": rows}, indent=1))
