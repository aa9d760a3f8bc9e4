#!/usr/bin/env python3
"""The three timed regions of SURVEY.md section 8(d) side by side (bench.py's `value` is region i+):
  (i+)  HBM-resident inputs -> HBM-resident outputs           plan.run
  (ii)  host buffers in -> host buffers out (PCIe inclusive)  ffx_score / set_pairs+run+fetch
  (iii) Ranking frame in -> Ranking frame out                 Index.__call__ + interpolate / rerank
"""
import argparse
import json
import sys
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

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=8_841_823)
ap.add_argument("--queries", type=int, default=6980)
ap.add_argument("--k", type=int, default=5000)
ap.add_argument("--api-rows", type=int, default=1_000_000)
ap.add_argument("--api-queries", type=int, default=200)
ap.add_argument("--api-k", type=int, default=1000)
ap.add_argument("--only-ii", action="store_true", help="stop after the host-to-host region")
args = ap.parse_args()
d, seed = 768, 2024
scale = float(1 / np.sqrt(d))
out = {}

idx = native.DeviceIndex(args.rows, d)
idx.fill_synthetic(seed, scale, 0, args.rows)
idx.set_doc_rows(np.arange(args.rows + 1, dtype=np.int64), np.arange(args.rows, dtype=np.int32))
idx.set_psg_rows(np.arange(args.rows, dtype=np.int32))
ids, sparse = B.candidates(args.rows, args.queries, args.k, seed)
pair_off = np.arange(args.queries + 1, dtype=np.int64) * args.k
queries = synth.rows(seed + 3, scale, 0, args.queries, d)
plan = native.Plan(idx)
plan.set_queries(queries)
plan.set_pairs(pair_off, ids, sparse, 2)
for _ in range(2):
    plan.run(0.1)
plan.sync()
t0 = time.perf_counter()
for _ in range(5):
    plan.run(0.1)
plan.sync()
out["i_device_resident_ms"] = (time.perf_counter() - t0) / 5 * 1e3
# (ii) persistent plan, host buffers each time
ts = []
for _ in range(4):
    t0 = time.perf_counter()
    plan.set_queries(queries)
    plan.set_pairs(pair_off, ids, sparse, 2)
    plan.run(0.1)
    ff, interp, _ = plan.fetch(interp=True)
    ts.append(time.perf_counter() - t0)
out["ii_host_to_host_persistent_plan_ms"] = min(ts) * 1e3
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    native.score(idx, queries, pair_off, ids, 2, sparse, 0.1)
    ts.append(time.perf_counter() - t0)
out["ii_host_to_host_one_shot_ffx_score_ms"] = min(ts) * 1e3
one_shot = native.score(idx, queries, pair_off, ids, 2, sparse, 0.1)
out["ii_one_shot_equals_plan"] = bool(np.array_equal(one_shot[0], ff) and np.array_equal(one_shot[1], interp))
out["ii_one_shot_device_ms_sum"] = one_shot[3]["ms_total"]
# the same call with the caller's buffers page-locked in place (ffx_host_pin), lock time included / excluded
ff_o, in_o = np.empty_like(ff), np.empty_like(interp)
t0 = time.perf_counter()
with native.pinned(queries, pair_off, ids, sparse, ff_o, in_o):
    t1 = time.perf_counter()
    ts = []
    for _ in range(3):
        t2 = time.perf_counter()
        native.score(idx, queries, pair_off, ids, 2, sparse, 0.1, out=(ff_o, in_o, None))
        ts.append(time.perf_counter() - t2)
out["ii_pin_in_place_ms"] = (t1 - t0) * 1e3
out["ii_host_to_host_pinned_in_place_ms"] = min(ts) * 1e3
# ... and with buffers that were allocated page-locked (ffx_host_alloc)
pq, po, pi_, ps = (native.pinned_empty(a.shape, a.dtype) for a in (queries, pair_off, ids, sparse))
for dst, src in ((pq, queries), (po, pair_off), (pi_, ids), (ps, sparse)):
    dst[...] = src
pf, pn = native.pinned_empty(ff.shape, np.float32), native.pinned_empty(ff.shape, np.float32)
ts = []
for _ in range(4):
    t0 = time.perf_counter()
    native.score(idx, pq, po, pi_, 2, ps, 0.1, out=(pf, pn, None))
    ts.append(time.perf_counter() - t0)
out["ii_host_to_host_pinned_alloc_ms"] = min(ts) * 1e3
out["ii_pinned_equals_plan"] = bool(np.array_equal(pf, ff) and np.array_equal(pn, interp))
out["ii_bytes_moved"] = int(queries.nbytes + ids.nbytes + sparse.nbytes + pair_off.nbytes + 2 * ff.nbytes)
out["queries"], out["k"] = args.queries, args.k
plan.close()
idx.close()
if args.only_ii:
    print(json.dumps(out, indent=1))
    sys.exit(0)

# (iii) the Python API with string ids, reduced size (host-bound)
n, nq, k = args.api_rows, args.api_queries, args.api_k
qv = synth.rows(seed + 3, scale, 0, nq, d)
index = InMemoryIndex(LambdaEncoder(lambda t: qv[int(t)]), mode=Mode.MAXP, init_size=n)
t0 = time.perf_counter()
step = 100_000
for lo in range(0, n, step):
    hi = min(n, lo + step)
    index.add(synth.rows(seed, scale, lo, hi - lo, d), doc_ids=[f"D{i}" for i in range(lo, hi)])
out["iii_add_s"] = time.perf_counter() - t0
cid, csp = B.candidates(n, nq, k, seed)
df = pd.DataFrame({"q_id": np.repeat([f"q{i:05d}" for i in range(nq)], k), "id": [f"D{i}" for i in cid], "score": csp})
t0 = time.perf_counter()
first = Ranking(df, queries={f"q{i:05d}": str(i) for i in range(nq)})
out["iii_ranking_ctor_s"] = time.perf_counter() - t0
index(first)  # warm
t0 = time.perf_counter()
dense = index(first)
out["iii_index_call_s"] = time.perf_counter() - t0
t0 = time.perf_counter()
final = first.interpolate(dense, 0.1)
out["iii_interpolate_s"] = time.perf_counter() - t0
t0 = time.perf_counter()
fused = index.rerank(first, 0.1)
out["iii_rerank_fused_s"] = time.perf_counter() - t0
# the same pipeline file to file: native parser + id resolution + device sort + native writer
import tempfile  # noqa: E402

with tempfile.TemporaryDirectory() as tmp:
    run_in, run_out, run_py = Path(tmp) / "in.tsv", Path(tmp) / "out.tsv", Path(tmp) / "py.tsv"
    first.save(run_in)
    queries_map = {f"q{i:05d}": str(i) for i in range(nq)}
    index.rerank_file(run_in, run_out, 0.1, queries_map)  # warm (builds the native id table once)
    t0 = time.perf_counter()
    index.rerank_file(run_in, run_out, 0.1, queries_map)
    out["iii_rerank_file_native_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    f2 = Ranking.from_file(run_in).attach_queries(queries_map)
    f2.interpolate(index(f2), 0.1).save(run_py)
    out["iii_readme_pipeline_python_s"] = time.perf_counter() - t0
    a, b = run_out.read_text().splitlines(), run_py.read_text().splitlines()
    out["iii_files_identical"] = a == b
    diff = [(x, y) for x, y in zip(a, b) if x != y]
    out["iii_lines"], out["iii_differing_lines"], out["iii_first_differences"] = (len(a), len(b)), len(diff), diff[:4]
out["iii_pairs"] = nq * k
out["iii_same_result"] = bool(fused == final)
out["iii_kernel_ms"] = index.last_stats["ms_gather_dot"]
print(json.dumps(out, indent=1))
