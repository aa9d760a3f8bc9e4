#!/usr/bin/env python3
"""Early stopping (reference index/base.py:316-387) at scale: how many look-ups
it saves and what that does to the wall time of Index.__call__."""
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
from fast_forward_amd import Ranking, synth  # noqa: E402
from fast_forward_amd.encoder import LambdaEncoder  # noqa: E402
from fast_forward_amd.index import InMemoryIndex, Mode  # noqa: E402
import bench as B  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1_000_000)
ap.add_argument("--queries", type=int, default=2000)
ap.add_argument("--k", type=int, default=1000)
ap.add_argument("--cutoff", type=int, default=10)
ap.add_argument("--alpha", type=float, default=0.2)
ap.add_argument("--depths", default="50,200,1000")
ap.add_argument("--profile", action="store_true")
args = ap.parse_args()
d, seed = 768, 2024
scale = float(1 / np.sqrt(d))
qv = synth.rows(seed + 3, scale, 0, args.queries, d)
index = InMemoryIndex(LambdaEncoder(lambda t: qv[int(t)]), mode=Mode.PASSAGE, init_size=args.rows)
for lo in range(0, args.rows, 200_000):
    hi = min(args.rows, lo + 200_000)
    index.add(synth.rows(seed, scale, lo, hi - lo, d), psg_ids=[f"P{i}" for i in range(lo, hi)])
ids, sparse = B.candidates(args.rows, args.queries, args.k, seed)
sparse = -np.sort(-sparse.reshape(args.queries, args.k), axis=1).reshape(-1)  # a first-stage run is sorted by score
df = pd.DataFrame({"q_id": np.repeat([f"q{i:05d}" for i in range(args.queries)], args.k), "id": [f"P{i}" for i in ids], "score": sparse})
first = Ranking(df, queries={f"q{i:05d}": str(i) for i in range(args.queries)})
depths = tuple(int(x) for x in args.depths.split(","))
out = {"queries": args.queries, "k": args.k, "cutoff": args.cutoff, "alpha": args.alpha, "depths": depths}
index(first)  # warm
t0 = time.perf_counter()
full = index(first)
out["full_s"] = time.perf_counter() - t0
out["full_lookups"] = args.queries * args.k
t0 = time.perf_counter()
es = index(first, early_stopping=args.cutoff, early_stopping_alpha=args.alpha, early_stopping_depths=depths)
out["early_stopping_first_call_s"] = time.perf_counter() - t0  # (sizes the plan's buffers)
t0 = time.perf_counter()
es = index(first, early_stopping=args.cutoff, early_stopping_alpha=args.alpha, early_stopping_depths=depths)
out["early_stopping_s"] = time.perf_counter() - t0
out["early_stopping_device_stats"] = {k: index.last_stats[k] for k in ("early_stopping_scored_pairs", "early_stopping_intervals", "ms_total")}
# the same call with the criterion evaluated on the host each interval (round 1's path: D2H of the interval's
# scores, lexsort + reduceat in NumPy, ids re-resolved and re-uploaded per interval)
from fast_forward_amd.index.api import Index  # noqa: E402
from fast_forward_amd.index.hbm import InMemoryIndex as _IMI  # noqa: E402

device_impl = _IMI._early_stopping_kept
_IMI._early_stopping_kept = Index._early_stopping_kept
t0 = time.perf_counter()
es_host = index(first, early_stopping=args.cutoff, early_stopping_alpha=args.alpha, early_stopping_depths=depths)
out["early_stopping_host_criterion_s"] = time.perf_counter() - t0
_IMI._early_stopping_kept = device_impl
out["host_and_device_criterion_agree"] = bool(es_host == es)
if "--profile" in sys.argv:
    import cProfile
    import pstats

    pr = cProfile.Profile()
    pr.enable()
    index(first, early_stopping=args.cutoff, early_stopping_alpha=args.alpha, early_stopping_depths=depths)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
out["early_stopping_lookups"] = int(len(es._df))
# the top-`cutoff` of the interpolated ranking must not change (that is the point of the criterion)
a = first.interpolate(full, args.alpha).cut(args.cutoff)
b = first.interpolate(es, args.alpha)  # unscored candidates count as dense 0 here, so compare on the scored ones
top_full = {q: set(a[q]) for q in a.q_ids}
b_scored = Ranking(b._df.merge(es._df[["q_id", "id"]], on=["q_id", "id"]), dtype=np.float32).cut(args.cutoff)
same = sum(top_full[q] == set(b_scored[q]) for q in top_full)
out["queries_with_identical_top_cutoff"] = same
print(json.dumps(out, indent=1))
