#!/usr/bin/env python3
"""Does the speed of gather_dot belong to the ALLOCATION?  K copies of the headline index (27 GB each) live side by side in one
process; the same batch is scored on each, round after round.  If a copy keeps its time from round to round while the copies
differ, the physical placement an allocation was handed decides (HISTORY section 5: +- 3 % between processes), and choosing among
a few allocations when an index is created would be worth its 30 ms.  Prints one JSON line."""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "fast-forward-indexes_amd"))
sys.path.insert(0, str(ROOT))
from fast_forward_amd import native, synth  # noqa: E402
import bench as B  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4
rows, d, nq, k = 8_841_823, 768, 6980, 5000
scale = float(1 / np.sqrt(d))
ids, sparse = B.candidates(rows, nq, k, 2024)
pair_off = np.arange(nq + 1, dtype=np.int64) * k
queries = synth.rows(2027, scale, 0, nq, d)
copies = []
for c in range(K):
    idx = native.DeviceIndex(rows, d)
    idx.fill_synthetic(2024, scale, 0, rows)
    idx.set_doc_rows(np.arange(rows + 1, dtype=np.int64), np.arange(rows, dtype=np.int32))
    idx.set_psg_rows(np.arange(rows, dtype=np.int32))
    plan = native.Plan(idx)
    plan.set_queries(queries)
    plan.set_pairs(pair_off, ids, sparse, 2)
    copies.append((idx, plan))
# a short probe beside the full step: 2 M candidates spread over the whole index (what an index could afford at creation)
probe_q = 400
table = {c: {"full_ms": [], "probe_ms": []} for c in range(K)}
probes = []
for idx, _ in copies:
    p = native.Plan(idx)
    p.set_queries(queries[:probe_q])
    p.set_pairs(pair_off[:probe_q + 1], ids[:probe_q * k], sparse[:probe_q * k], 2)
    probes.append(p)
for rnd in range(4):
    for c, (idx, plan) in enumerate(copies):
        ms = []
        for _ in range(4):
            plan.run(0.1, exact_sum=True, sort=True)
            plan.sync()
            ms.append(plan.stats()["ms_gather_dot"])
        table[c]["full_ms"].append(round(min(ms[1:]), 3))
        ms = []
        for _ in range(4):
            probes[c].run(0.1, exact_sum=True, sort=False)
            probes[c].sync()
            ms.append(probes[c].stats()["ms_gather_dot"])
        table[c]["probe_ms"].append(round(min(ms[1:]), 4))
full = np.array([table[c]["full_ms"] for c in range(K)])
probe = np.array([table[c]["probe_ms"] for c in range(K)])
def full_ms(plan):
    ms = []
    for _ in range(4):
        plan.run(0.1, exact_sum=True, sort=True)
        plan.sync()
        ms.append(plan.stats()["ms_gather_dot"])
    return round(min(ms[1:]), 3)


# does a copy keep its figure when the OTHERS go?  (ffx_index_reseat frees the losers: tools/reseat_check.py saw the block that stayed
# slow down by 2 - 3 %.)  The fastest copy is scored with all alive, with the others closed one by one, and with new blocks allocated again
aftermath = {}
if "--aftermath" in sys.argv:
    best = int(np.argmin(full.min(axis=1)))
    keep_idx, keep_plan = copies[best]
    aftermath["copy"] = best
    aftermath["all_alive_ms"] = full_ms(keep_plan)
    aftermath["as_the_others_close_ms"] = []
    for c, (idx, plan) in enumerate(copies):
        if c == best:
            continue
        probes[c].close()
        plan.close()
        idx.close()
        aftermath["as_the_others_close_ms"].append(full_ms(keep_plan))
    fresh = [native.DeviceIndex(rows, d) for _ in range(3)]
    aftermath["three_new_blocks_alive_ms"] = full_ms(keep_plan)
    for f in fresh:
        f.fill_synthetic(1, scale, 0, rows)
    aftermath["... and written_ms"] = full_ms(keep_plan)
    for f in fresh:
        f.close()
    aftermath["... and closed_ms"] = full_ms(keep_plan)
    copies = [(keep_idx, keep_plan)]
    K_alive = [best]
else:
    K_alive = list(range(K))
for c, (idx, _) in zip(K_alive, copies):
    table[c]["reseat_probe_ms"] = round(min(idx.probe_placement() for _ in range(2)), 4)  # what ffx_index_reseat would compare
    table[c]["device_ptr"] = hex(idx.device_ptr() if callable(idx.device_ptr) else idx.device_ptr)
print(json.dumps({"copies": K, "index_GB": round(rows * d * 4 / 1e9, 2), "gather_dot_ms_by_copy_and_round": table, "aftermath": aftermath,
                  "spread_between_copies_pct": round(float((full.min(axis=1).max() / full.min(axis=1).min() - 1) * 100), 2),
                  "spread_within_a_copy_pct": round(float(((full.max(axis=1) / full.min(axis=1)).max() - 1) * 100), 2),
                  "probe_ranks_copies_like_the_full_step": bool((np.argsort(probe.min(axis=1)) == np.argsort(full.min(axis=1))).all()),
                  "order_by_full_step": np.argsort(full.min(axis=1)).tolist(), "order_by_short_step": np.argsort(probe.min(axis=1)).tolist(),
                  "order_by_reseat_probe": np.argsort([table[c].get("reseat_probe_ms", 0.0) for c in range(K)]).tolist()}))
