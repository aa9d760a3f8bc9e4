#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs of tools/profile_r03.sh into profiles/r03/: the bench's kernel stats / trace, its HBM traffic
(FETCH_SIZE x 2 on gfx950 + WRITE_SIZE), and per row length d the traffic of ONE scoring pass (all of its gather
launches: a long row is one launch per segment) against the algorithmic bytes."""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

src, dst = Path(sys.argv[1]), Path(sys.argv[2])
dst.mkdir(parents=True, exist_ok=True)


def dispatches(folder: str) -> list[dict]:
    """One dict per kernel dispatch of a --pmc pass (dispatch order): name + counters."""
    per = defaultdict(dict)
    for f in (src / folder).rglob("*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            key = (f.name, int(row["Dispatch_Id"]))
            per[key]["name"] = row["Kernel_Name"]
            per[key][row["Counter_Name"]] = per[key].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    return [per[k] for k in sorted(per)]


def short(name: str) -> str:
    return name.replace("void ffx::", "").split("(")[0]


summary = {}
for f in (src / "bench_trace").rglob("*kernel_stats.csv"):
    (dst / "bench_kernel_stats.csv").write_text(f.read_text())
    for row in csv.DictReader(open(f)):
        if "gather_dot" in row["Name"]:
            summary["bench_gather_dot_trace"] = {"kernel": short(row["Name"]), "calls": int(row["Calls"]), "avg_ms": float(row["AverageNs"]) / 1e6,
                                                "min_ms": float(row["MinNs"]) / 1e6, "max_ms": float(row["MaxNs"]) / 1e6}
for f in (src / "bench_trace").rglob("*kernel_trace.csv"):
    rows = list(csv.DictReader(open(f)))
    keep = [r for r in rows if "gather_dot" in r["Kernel_Name"]][-20:]
    with open(dst / "bench_kernel_trace_gather_dot_last20.csv", "w", newline="") as g:
        w = csv.DictWriter(g, fieldnames=list(keep[0].keys()))
        w.writeheader()
        w.writerows(keep)
for ln in (src / "bench_trace.stdout").read_text().strip().splitlines():
    if ln.startswith("{"):
        b = json.loads(ln)
        summary["bench_line_of_the_traced_run"] = {"value": b["value"], "ms_per_step": b["ms_per_step"], "kernel_ms_avg_hip_events": b["roofline"]["kernel_ms_avg"],
                                                   "achieved_GBs": b["roofline"]["achieved"], "frac": b["roofline"]["frac"],
                                                   "algorithmic_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_launch"]}
        (dst / "bench_traced.json").write_text(ln + "\n")
# the bench's gather_dot: traffic per launch
bench = {}
for counter, folder in (("FETCH_SIZE", "bench_fetch"), ("WRITE_SIZE", "bench_write")):
    vals = [d[counter] for d in dispatches(folder) if "gather_dot" in d["name"] and counter in d]
    if vals:
        bench[counter + "_KiB_avg"] = sum(vals) / len(vals)
        bench["launches"] = len(vals)
tcc = [d for d in dispatches("bench_tcc") if "gather_dot" in d["name"]]
if tcc:
    h, m = sum(d.get("TCC_HIT_sum", 0) for d in tcc), sum(d.get("TCC_MISS_sum", 0) for d in tcc)
    bench["L2_hit_rate"] = h / max(1.0, h + m)
if "FETCH_SIZE_KiB_avg" in bench:
    bench["fetch_bytes_x2_per_launch"] = 2 * 1024 * bench["FETCH_SIZE_KiB_avg"]
    bench["write_bytes_per_launch"] = 1024 * bench.get("WRITE_SIZE_KiB_avg", 0.0)
    alg = summary.get("bench_line_of_the_traced_run", {}).get("algorithmic_bytes_per_launch")
    if alg:
        bench["traffic_over_algorithmic"] = (bench["fetch_bytes_x2_per_launch"] + bench["write_bytes_per_launch"]) / alg
summary["bench_pmc"] = bench
# per row length: the gather launches of one pass
dims = {}
for d in (1100, 2000, 5000, 100, 200, 300):
    out_line = None
    for ln in (src / f"dims{d}_fetch.stdout").read_text().splitlines() if (src / f"dims{d}_fetch.stdout").exists() else []:
        pass
    try:
        res = json.loads((src / f"dims{d}_fetch.stdout").read_text())[str(d)]
    except (OSError, ValueError, KeyError):
        continue
    gathers = [x for x in dispatches(f"dims{d}_fetch") if "gather_dot" in x["name"]]
    combines = [x for x in dispatches(f"dims{d}_fetch") if "combine_tree" in x["name"]]
    passes = 3
    per_pass = len(gathers) // passes
    fetch = 2 * 1024 * sum(x["FETCH_SIZE"] for x in gathers) / passes
    fetch_combine = 2 * 1024 * sum(x["FETCH_SIZE"] for x in combines) / passes if combines else 0.0
    e = {"kernel": short(gathers[0]["name"]), "gather_launches_per_pass": per_pass, "variant": res["anyd"]["variant"], "GBs_algorithmic": res["anyd"]["GBs"],
         "algorithmic_bytes_per_pass": res["algorithmic_bytes_per_launch"], "fetch_bytes_x2_per_pass": fetch, "combine_kernel_fetch_bytes_x2_per_pass": fetch_combine,
         "fetch_over_algorithmic": (fetch + fetch_combine) / res["algorithmic_bytes_per_launch"]}
    t = [x for x in dispatches(f"dims{d}_tcc") if "gather_dot" in x["name"]]
    if t:
        h, m = sum(x.get("TCC_HIT_sum", 0) for x in t), sum(x.get("TCC_MISS_sum", 0) for x in t)
        e["L2_hit_rate"] = h / max(1.0, h + m)
    row_bytes = d * 4
    e["lines_of_128B_per_row_min"] = -(-row_bytes // 128)
    dims[str(d)] = e
summary["row_lengths (tools/dims_bench.py <d> --only-shipping: 2000 queries x k=5000, index ~13 GB; 3 passes)"] = dims
summary["note"] = ("FETCH_SIZE is in KiB and reads 1/2 of the bytes on gfx950 (MI355X_MICROARCH.md, HBM section; calibrated in profiles/r01/pmc_summary.json): "
                   "x2 applied.  Every --pmc pass is its own run, never combined with tracing.")
(dst / "pmc_summary.json").write_text(json.dumps(summary, indent=1) + "\n")
print(json.dumps(summary, indent=1)[:5000])
