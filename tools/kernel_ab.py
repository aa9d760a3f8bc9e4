#!/usr/bin/env python3
"""Interleaved A/B of gather_dot instantiations in ONE process on ONE box (box-to-box spread is
larger than the effects being measured).  Variants are selected with FFX_DEBUG_VARIANT=gather=N
(csrc/ffx.hip, launch_gather_dot); 0 = the shipping kernel for the chosen --sum."""
import argparse
import json
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "fast-forward-indexes_amd"))
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import _experiments  # noqa: E402

_experiments.use()  # the switches this tool flips exist in the experiments build only
from fast_forward_amd import native, synth  # noqa: E402
import bench as B  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=8_841_823)
ap.add_argument("--queries", type=int, default=6980)
ap.add_argument("--k", type=int, default=5000)
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--variants", default="0,900,907,911,913,916,919,920,fast,951,953,955,958,959")
ap.add_argument("--store", default="f32", choices=["f32", "f16"])
ap.add_argument("--dim", type=int, default=768)
args = ap.parse_args()

d, seed = args.dim, 2024
scale = float(1 / np.sqrt(d))
store = np.float16 if args.store == "f16" else np.float32
idx = native.DeviceIndex(args.rows, d, dtype=store)
idx.fill_synthetic(seed, scale, 0, args.rows)
idx.set_doc_rows(np.arange(args.rows + 1, dtype=np.int64), np.arange(args.rows, dtype=np.int32))
idx.set_psg_rows(np.arange(args.rows, dtype=np.int32))
ids, sparse = B.candidates(args.rows, args.queries, args.k, seed)
pair_off = np.arange(args.queries + 1, dtype=np.int64) * args.k
plan = native.Plan(idx)
plan.set_queries(synth.rows(seed + 3, scale, 0, args.queries, d))
plan.set_pairs(pair_off, ids, sparse, 2)

variants = args.variants.split(",")
ref = None
times = {v: [] for v in variants}
for rnd in range(args.rounds + 1):
    for v in variants:
        exact = True
        if v == "fast":
            os.environ["FFX_DEBUG_VARIANT"] = "gather=0"
            exact = False
        else:
            os.environ["FFX_DEBUG_VARIANT"] = f"gather={v}"
            exact = not (v.startswith("95"))
        if v == "strided":  # small-row dims: the strided-load form
            os.environ["FFX_DEBUG_VARIANT"] = "gather=800"
        plan.run(0.1, exact_sum=exact)
        st = plan.stats()
        if rnd == 0:  # warm-up round doubles as a correctness check of every exact variant
            ff, _, _ = plan.fetch()
            if ref is None:
                ref = ff
            elif exact:
                assert np.array_equal(ff.view(np.uint32), ref.view(np.uint32)), f"variant {v} changes the scores"
            else:
                assert np.abs(ff - ref).max() < 1e-4, f"variant {v} off by {np.abs(ff - ref).max()}"
            continue
        times[v].append(st["ms_gather_dot"])
rows = args.queries * args.k
row_bytes = d * np.dtype(store).itemsize + 4
out = {v: {"ms_mean": float(np.mean(t)), "ms_min": float(np.min(t)), "GBs_mean": rows * row_bytes / np.mean(t) / 1e6} for v, t in times.items()}
for v, r in sorted(out.items(), key=lambda kv: kv[1]["ms_mean"]):
    print(f"{v:>5}  mean {r['ms_mean']:.3f} ms  min {r['ms_min']:.3f} ms  {r['GBs_mean']:.0f} GB/s")
print(json.dumps(out))
