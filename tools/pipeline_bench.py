#!/usr/bin/env python3
"""The reference's documented pipeline, file to file, through the Python API at scale:

    first = Ranking.from_file(run, queries).cut(k);  dense = index(first);  first.interpolate(dense, a).save(out)

timed stage by stage on an 8.8 M x 768 index, with large runs read into the coded form (default) and into a
pandas frame (the route of small files), next to Index.rerank_file (no Ranking in between).  Outputs must be equal."""
import argparse
import filecmp
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "fast-forward-indexes_amd"))
sys.path.insert(0, str(ROOT))
from fast_forward_amd import Ranking, native, synth  # noqa: E402
from fast_forward_amd.encoder import ArrayEncoder  # noqa: E402
from fast_forward_amd.index import InMemoryIndex, Mode  # noqa: E402
import fast_forward_amd.ranking as R  # noqa: E402
import bench as B  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=8_841_823)
ap.add_argument("--queries", type=int, default=2000)
ap.add_argument("--k", type=int, default=5000)
ap.add_argument("--dir", default="/tmp/ffx_pipeline")
ap.add_argument("--profile", action="store_true", help="cProfile of from_file + index(first) (coded route) on stderr")
args = ap.parse_args()
d, seed = 768, 2024
scale = float(1 / np.sqrt(d))
work = Path(args.dir)
work.mkdir(parents=True, exist_ok=True)

store = native.DeviceIndex(args.rows, d)
store.fill_synthetic(seed, scale, 0, args.rows)
doc_names = np.char.add("D", np.arange(args.rows).astype("U"))
index = InMemoryIndex.adopt(store, doc_ids=doc_names, doc_off=np.arange(args.rows + 1), mode=Mode.MAXP)
qvecs = synth.rows(seed + 3, scale, 0, args.queries, d)
q_names = [f"q{i}" for i in range(args.queries)]
index.query_encoder = ArrayEncoder(dict(zip((f"text {i}" for i in range(args.queries)), qvecs)))
queries = {q: f"text {i}" for i, q in enumerate(q_names)}

# the first-stage run file (written natively: codes + names)
ids, sparse = B.candidates(args.rows, args.queries, args.k, seed)
sparse = np.sort(sparse.reshape(args.queries, args.k), axis=1)[:, ::-1].reshape(-1).copy()
pair_off = np.arange(args.queries + 1, dtype=np.int64) * args.k
run_in = work / "first.tsv"
assert native.save_run_coded(run_in, q_names, pair_off, ids, doc_names.astype(object), sparse, "bm25")
out = {"rows": int(pair_off[-1]), "file_MB": run_in.stat().st_size / 1e6}


def pipeline(tag: str) -> dict:
    t = [time.perf_counter()]
    first = Ranking.from_file(run_in, queries); t.append(time.perf_counter())
    first = first.cut(args.k); t.append(time.perf_counter())
    dense = index(first); t.append(time.perf_counter())
    final = first.interpolate(dense, 0.1); t.append(time.perf_counter())
    final.save(work / f"out_{tag}.tsv"); t.append(time.perf_counter())
    names = ("from_file_s", "cut_s", "call_s", "interpolate_s", "save_s")
    res = {n: t[i + 1] - t[i] for i, n in enumerate(names)}
    res["total_s"] = t[-1] - t[0]
    return res


pipeline("warm")  # first call: id resolution caches, page-locked result block
if args.profile:
    import cProfile
    import io
    import pstats

    pr = cProfile.Profile()
    pr.enable()
    first = Ranking.from_file(run_in, queries)
    dense = index(first)
    pr.disable()
    buf = io.StringIO()
    pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(14)
    print(buf.getvalue(), file=sys.stderr)
out["coded"] = pipeline("coded")
keep = R._CODED_FROM_FILE_ROWS
R._CODED_FROM_FILE_ROWS = 10**15
out["frame"] = pipeline("frame")
R._CODED_FROM_FILE_ROWS = keep
t0 = time.perf_counter()
index.rerank_file(run_in, work / "out_rerank_file.tsv", 0.1, queries)
out["rerank_file_s"] = time.perf_counter() - t0
# ... and its stages
table = index._native_id_table(index.mode)
t0 = time.perf_counter()
run = native.RunFile(run_in, table)
t1 = time.perf_counter()
q_ids = run.query_ids()
r_off, r_ord, r_sparse = run.export(0)
t2 = time.perf_counter()
qvec = np.ascontiguousarray(index.encode_queries([queries[q] for q in q_ids]))
_, interp, order = index._score_pairs(qvec, r_off, r_ord, index.mode, r_sparse, 0.1, sort=True)
t3 = time.perf_counter()
run.write(work / "out_stages.tsv", interp, order, tag=None)
t4 = time.perf_counter()
run.close()
t5 = time.perf_counter()
plain = native.RunFile(run_in, None)
t6 = time.perf_counter()
plain.close()
out["rerank_file_stages_s"] = {"parse_and_resolve": t1 - t0, "export": t2 - t1, "encode_and_score": t3 - t2, "write": t4 - t3, "parse_without_table": t6 - t5}
out["coded_equals_frame"] = filecmp.cmp(work / "out_coded.tsv", work / "out_frame.tsv", shallow=False)
a = (work / "out_coded.tsv").read_bytes().splitlines()
b = (work / "out_rerank_file.tsv").read_bytes().splitlines()
out["lines_differing_from_rerank_file"] = sum(x != y for x, y in zip(a, b)) + abs(len(a) - len(b))
print(json.dumps(out, indent=1))
