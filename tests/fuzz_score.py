#!/usr/bin/env python3
"""Randomised end-to-end check of the C ABI against the oracle: ragged batches, every mode, several
row lengths and stores, random slice sizes of ffx_score, sort on and off.  Not collected by pytest
(it is a soak: `python tests/fuzz_score.py --rounds 300` on the GPU box, under a minute); prints the
seed of any failure.  Last runs of round 1: 300 + 1000 configurations (seeds 0 and 1), all bit-exact;
round 2 (any-d kernel, in-LDS sort): 600 + 600 configurations (seeds 2 and 3), all bit-exact; round 2 later
(ffx_score's single-submission path for small single-row batches; slices through three plans with asynchronous
copies into page-locked results or helper threads into ordinary ones): seeds 4 and 5, see HISTORY.md; round 3 (segments
for long rows, uploads of all slices on one stream): 600 + 600 configurations (seeds 8 and 9), all bit-exact; round 5 (staged
transfers of ordinary memory, four plans, rows derived inside gather_dot; one configuration in eight LARGE): seeds 10 and 11; 12 - 17 after the
one-row derive, the query hint, the fast-order derive, graded staging and ffx_index_reseat went in (320 configurations each)."""
import argparse
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT / "fast-forward-indexes_amd", ROOT / "oracle"):
    sys.path.insert(0, str(p))
import ffx_oracle as fo  # noqa: E402  (a checking tool, like the tests)
from fast_forward_amd import native, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=150)
ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
bits = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)  # noqa: E731
DIMS = [8, 24, 64, 100, 128, 256, 300, 384, 512, 768, 1024, 1536, 2048,
        9, 17, 130, 200, 260, 500, 520, 700, 900, 1000, 1016, 1023, 1100, 5]  # second line (round 2): the any-d kernel's lengths, and its fall-backs
for rnd in range(args.rounds):
    seed = args.seed * 100_000 + rnd
    rng = np.random.default_rng(seed)
    # one configuration in eight is LARGE (round 5): candidate arrays of 1.5 - 5 MB in ordinary or page-locked memory, cut into
    # several slices -- the staged transfers of csrc/host_xfer.h, the four rotating plans and the result arrays leaving in halves
    large = rng.random() < 0.125
    d = int(rng.choice([64, 128, 768, 1024] if large else DIMS))
    store = np.float16 if rng.random() < 0.3 else np.float32
    n_docs = int(rng.integers(40_000, 80_000) if d > 128 else rng.integers(60_000, 120_000)) if large else int(rng.integers(1, 400))
    per_doc = rng.integers(1, 3 if large else 6, size=n_docs) if rng.random() < 0.7 else np.ones(n_docs, dtype=np.int64)  # single-row documents: the small path
    doc_off = np.concatenate([[0], np.cumsum(per_doc)]).astype(np.int64)
    n = int(doc_off[-1])
    doc_rows = rng.permutation(n).astype(np.int32)
    psg_row = rng.permutation(n).astype(np.int32)
    vectors = synth.rows(seed, 0.2, 0, n, d).astype(store)
    n_q = int(rng.integers(6, 14)) if large else int(rng.integers(1, 12))
    queries = synth.rows(seed + 1, 0.2, 0, n_q, d)
    mode = int(rng.integers(1, 5))
    pool = n if mode == 1 else n_docs
    lens = ([int(rng.integers(pool // 2, pool + 1)) for _ in range(n_q)] if large else
            [int(rng.integers(0, min(pool, 300 if rng.random() < 0.8 else 3000) + 1)) for _ in range(n_q)])
    pair_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pair_ids = np.concatenate([rng.choice(pool, size=k, replace=False) for k in lens] + [np.zeros(0, dtype=np.int64)]).astype(np.int32)
    sparse = (rng.integers(0, 5, size=len(pair_ids)) if rng.random() < 0.5 else rng.random(len(pair_ids)) * 10).astype(np.float32)
    alpha = float(rng.choice([0.0, 0.05, 0.3, 1.0]))
    variant = [f"score_slice_pairs={int(rng.choice([70_000, 300_000, 10**9] if large else [1, 7, 100, 10**9]))}"]
    if rng.random() < 0.5:
        variant.append("no_small_path")
    os.environ["FFX_DEBUG_VARIANT"] = ",".join(variant)
    out = None
    if rng.random() < 0.5:  # page-locked result arrays: results leave asynchronously on the out-stream
        out = tuple(native.pinned_empty(max(1, len(pair_ids)), t) for t in (np.float32, np.float32, np.int32))
    idx = native.DeviceIndex(n, d, dtype=store)
    idx.upload(vectors, 0)
    idx.set_doc_rows(doc_off, doc_rows)
    idx.set_psg_rows(psg_row)
    try:
        ff, interp, order, st = native.score(idx, queries, pair_off, pair_ids, mode, sparse, alpha, exact_sum=True, sort=True, out=out)
        want_ff, want_in = fo.score(vectors.astype(np.float32), doc_off, doc_rows, psg_row, queries, pair_off, pair_ids, mode, sparse, alpha)
        assert np.array_equal(bits(ff), bits(want_ff)), "dense"
        assert np.array_equal(bits(interp), bits(want_in)), "interp"
        assert np.array_equal(order.astype(np.int64), fo.order(want_in, pair_off)), "order"
        fast, _, _, _ = native.score(idx, queries, pair_off, pair_ids, mode, exact_sum=False)
        if len(fast):
            assert np.abs(fast - want_ff).max() <= 1e-4 * max(1.0, float(np.abs(want_ff).max())), "fast"
    except AssertionError as e:
        print(f"FAIL seed={seed} d={d} store={np.dtype(store).name} mode={mode} n_q={n_q} pairs={len(pair_ids)} what={e}")
        sys.exit(1)
    finally:
        idx.close()
    if rnd % 25 == 0:
        print(f"round {rnd}: d={d} {np.dtype(store).name} mode={mode} pairs={len(pair_ids)} variant={st['kernel_variant']} ok", flush=True)
print(f"fuzz ok: {args.rounds} configurations")
