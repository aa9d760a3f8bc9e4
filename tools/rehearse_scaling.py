#!/usr/bin/env python3
"""Predict the N-GPU scaling of the row-sharded bench from ONE GPU.

For N in --ranks, `bench.py --as-rank-of N [--as-rank r]` rehearses one rank of an N-rank run on the
one GPU (its shard = rows/N, its local CSR ~ P/N pairs, its owned block ~ Q/N queries, the exchange
moving the same bytes through this GPU's memory).  From the per-rank step times this prints

    N | step ms | ideal ms (gather_dot of P/N rows alone) | step / ideal | predicted q/s | predicted efficiency

where predicted efficiency = step(1) / (N * step(N)): what the scaling would be if the fabric added
nothing to the step (the exchange of batch i overlaps the scoring of batch i+1 by construction).
Writes the table as JSON (--out).  Not a measurement of N GPUs: the driver's 8-GPU node does that.
"""

from __future__ import annotations

import argparse
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--which", default="first", choices=["first", "last", "both"], help="which rank of N to rehearse")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--out", default=None)
    ap.add_argument("rest", nargs="*", help="further bench.py arguments (after --)")
    args = ap.parse_args()
    rows = []
    for n in [int(x) for x in args.ranks.split(",")]:
        for r in sorted({0} if args.which == "first" else {n - 1} if args.which == "last" else {0, n - 1}):
            cmd = [sys.executable, str(ROOT / "bench.py"), "--as-rank-of", str(n), "--as-rank", str(r), "--steps", str(args.steps),
                   "--warmup", str(args.warmup), "--cpu-sample-queries", "16", "--cpu-seconds", "0.2"] + args.rest
            out = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, check=True).stdout
            line = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
            reh = line["rehearsal"]
            rows.append({"N": n, "rank": r, "step_ms": reh["step_ms"], "ideal_ms": reh["ideal_ms"], "step_over_ideal": reh["step_over_ideal"],
                         "local_pairs": reh["local_pairs"], "owned_pairs": reh["owned_pairs"],
                         "exchange_ms": line["exchange_ms_per_rank"][r], "merge_ms": line["merge_ms_per_rank"][r],
                         "local_step_ms": line["local_step_ms_per_rank"][r], "parity": line.get("parity_on_cpu_sample"),
                         "set_once_route_and_upload_s": line["route_and_upload_next_batch_s"], "host_issue_ms_per_step": line.get("host_issue_ms_per_step"),
                         "set_once_host_to_host_batch_ms": line.get("set_once_host_to_host_batch_ms"),
                         # a FRESH batch per step, every rank handed only the queries it owns (ffx_sharded_stage_owned)
                         "fresh_batch_step_ms": line["fresh_batch_step_ms"], "fresh_over_ideal": line["fresh_batch_step_ms"] / reh["ideal_ms"],
                         "fresh_query_vectors": line.get("fresh_batch_query_vectors"), "stage_h2d_bytes": line["stage_h2d_bytes_per_rank"][r],
                         "stage_route_sent_bytes": line["stage_route_sent_bytes_per_rank"][r], "stage_device_ms": line["stage_device_ms_per_rank"][r],
                         "fresh_gather_dot_ms": line["fresh_batch_gather_dot_ms_per_rank"][r], "rehearsal_exchange": line.get("rehearsal_exchange"),
                         "queries": line["config"]["queries"]})
            print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
    base = next((x["step_ms"] for x in rows if x["N"] == 1), None)
    base_fresh = next((x["fresh_batch_step_ms"] for x in rows if x["N"] == 1), None)
    for x in rows:
        x["predicted_queries_per_s"] = x["queries"] / (x["step_ms"] * 1e-3)
        x["predicted_efficiency"] = None if base is None else base / (x["N"] * x["step_ms"])
        x["fresh_predicted_queries_per_s"] = x["queries"] / (x["fresh_batch_step_ms"] * 1e-3)
        x["fresh_predicted_efficiency"] = None if base_fresh is None else base_fresh / (x["N"] * x["fresh_batch_step_ms"])
    print(f"{'N':>3} {'rank':>4} {'step ms':>9} {'ideal ms':>9} {'step/ideal':>10} {'fresh ms':>9} {'fresh/ideal':>11} {'stage ms':>9} {'H2D MB':>8} {'pred eff':>8} {'fresh eff':>9}")
    for x in rows:
        eff = "-" if x["predicted_efficiency"] is None else f"{x['predicted_efficiency']:.3f}"
        feff = "-" if x["fresh_predicted_efficiency"] is None else f"{x['fresh_predicted_efficiency']:.3f}"
        print(f"{x['N']:>3} {x['rank']:>4} {x['step_ms']:>9.3f} {x['ideal_ms']:>9.3f} {x['step_over_ideal']:>10.3f} {x['fresh_batch_step_ms']:>9.3f} "
              f"{x['fresh_over_ideal']:>11.3f} {x['stage_device_ms']:>9.3f} {x['stage_h2d_bytes'] / 1e6:>8.1f} {eff:>8} {feff:>9}")
    if args.out:
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(json.dumps({"what": "single-GPU rehearsal of one rank of N (bench.py --as-rank-of N): predicted scaling, not measured",
                                              "rows": rows}, indent=1) + "\n")


if __name__ == "__main__":
    main()
