#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs of tools/profile_r02.sh into profiles/r02/*.json|csv (per kernel: launches,
average duration, FETCH_SIZE x2 / WRITE_SIZE bytes per launch, L2 hit rate)."""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

src, dst = Path(sys.argv[1]), Path(sys.argv[2])
dst.mkdir(parents=True, exist_ok=True)


def counters(folder: str) -> dict:
    """{kernel: {counter: [value per dispatch]}} for the gather/adc/lut kernels of one --pmc pass."""
    out = defaultdict(lambda: defaultdict(list))
    for f in (src / folder).rglob("*counter_collection.csv"):
        per_dispatch = defaultdict(dict)
        for row in csv.DictReader(open(f)):
            per_dispatch[(row["Dispatch_Id"], row["Kernel_Name"])][row["Counter_Name"]] = float(row["Counter_Value"])
        for (_, name), vals in per_dispatch.items():
            if any(k in name for k in ("gather_dot", "pq_adc", "pq_lut")):
                for c, v in vals.items():
                    out[name][c].append(v)
    return out


def short(name: str) -> str:
    return name.replace("void ffx::", "").split("(")[0]


summary = {}
# ---- kernel trace / stats of the bench command
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
line = (src / "bench_trace.stdout").read_text().strip().splitlines()
for ln in line:
    if ln.startswith("{"):
        b = json.loads(ln)
        summary["bench_line_of_the_traced_run"] = {"value": b["value"], "ms_per_step": b["ms_per_step"], "kernel_ms_avg_hip_events": b["roofline"]["kernel_ms_avg"],
                                                   "achieved_GBs": b["roofline"]["achieved"], "frac": b["roofline"]["frac"],
                                                   "algorithmic_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_launch"]}
        (dst / "bench_traced.json").write_text(ln + "\n")


def traffic(prefix: str, algorithmic: dict | None = None) -> dict:
    out = {}
    fetch = counters(f"{prefix}_fetch")
    write = counters(f"{prefix}_write") if (src / f"{prefix}_write").exists() else {}
    tcc = counters(f"{prefix}_tcc") if (src / f"{prefix}_tcc").exists() else {}
    lds = counters(f"{prefix}_lds") if (src / f"{prefix}_lds").exists() else {}
    for name, c in fetch.items():
        vals = c["FETCH_SIZE"]
        e = {"launches": len(vals), "FETCH_SIZE_KiB_avg": sum(vals) / len(vals), "fetch_bytes_x2_per_launch": 2 * 1024 * sum(vals) / len(vals)}
        if name in write:
            w = write[name]["WRITE_SIZE"]
            e["write_bytes_per_launch"] = 1024 * sum(w) / len(w)
        if name in tcc:
            h, m, r = (sum(tcc[name][k]) for k in ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"))
            e["L2_hit_rate"] = h / max(1.0, h + m)
            e["L2_requests_per_launch"] = r / len(tcc[name]["TCC_REQ_sum"])
        if name in lds:
            e["LDS"] = {k: sum(v) / len(v) for k, v in lds[name].items()}
        out[short(name)] = e
    return out


summary["bench_pmc"] = traffic("bench")
summary["anyd_kernel_pmc (tools/dims_bench.py 100,300,1000: 2000 queries x k=5000 per d; the first 6 launches of a kernel are the any-d kernel, the next 6 the staged one)"] = traffic("dims")
summary["pq_decode_then_dot_pmc (bench.py --store pq, FFX_PQ_NO_ADC=1)"] = traffic("pq_decode")
summary["note"] = ("FETCH_SIZE is in KiB and reads 1/2 of the bytes on gfx950 (MI355X_MICROARCH.md, HBM section; calibrated for these access patterns in "
                   "profiles/r01/pmc_summary.json): x2 applied.  Every --pmc pass is its own run, never combined with tracing.")
(dst / "pmc_summary.json").write_text(json.dumps(summary, indent=1) + "\n")
print(json.dumps(summary, indent=1)[:6000])
