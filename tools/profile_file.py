#!/usr/bin/env python3
"""Stage times of Index.rerank_file (native TREC parse -> score -> native write)."""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import pandas as pd

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "fast-forward-indexes_amd"))
sys.path.insert(0, str(ROOT))
from fast_forward_amd import Ranking, native, synth  # noqa: E402
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
with tempfile.TemporaryDirectory() as tmp:
    run_in, run_out = Path(tmp) / "in.tsv", Path(tmp) / "out.tsv"
    first.save(run_in)
    queries = {f"q{i:05d}": str(i) for i in range(nq)}
    index.rerank_file(run_in, run_out, 0.1, queries)
    table = index._native_id_table(index.mode)
    for _ in range(2):
        t0 = time.perf_counter()
        run = native.RunFile(run_in, table)
        t1 = time.perf_counter()
        q_ids = run.query_ids()
        pair_off, ordinals, sparse = run.export(0)
        t2 = time.perf_counter()
        qvec = np.ascontiguousarray(index.encode_queries([queries[q] for q in q_ids]))
        _, interp, order = index._score_pairs(qvec, pair_off, ordinals, index.mode, sparse, 0.1, sort=True)
        t3 = time.perf_counter()
        run.write(run_out, interp, order, tag=None)
        t4 = time.perf_counter()
        run.close()
        print(f"parse {t1 - t0:.3f}  export {t2 - t1:.3f}  encode+score {t3 - t2:.3f}  write {t4 - t3:.3f}  total {t4 - t0:.3f}")
    print("file MB", run_in.stat().st_size / 1e6)
