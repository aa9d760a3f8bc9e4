#!/usr/bin/env python3
"""Post-process a `rocprofv3 --hip-trace --memory-copy-trace --kernel-trace --output-format csv` directory: list every
HIP API call longer than --ms milliseconds with what the device did meanwhile (copies and kernels that overlap the call).
Used to look at the 10 - 35 ms wait the second and third streamed call of a process show in their first upload."""
import argparse
import csv
import glob
import json
import os

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--ms", type=float, default=5.0)
ap.add_argument("--skip", default="hipHostRegister,hipHostMalloc,hipMalloc,hipHostFree,hipFree,hipModuleLoad,hipDeviceSynchronize")
args = ap.parse_args()


def rows(pattern):
    out = []
    for path in glob.glob(os.path.join(args.dir, "**", pattern), recursive=True):
        with open(path, newline="") as f:
            out.extend(csv.DictReader(f))
    return out


def span(r):
    return int(r["Start_Timestamp"]), int(r["End_Timestamp"])


api, copies, kernels = rows("*hip_api_trace.csv"), rows("*memory_copy_trace.csv"), rows("*kernel_trace.csv")
skip = set(args.skip.split(","))
t0 = min(span(r)[0] for r in api) if api else 0
report = {"api_calls": len(api), "copies": len(copies), "kernels": len(kernels), "long_calls": []}
for r in api:
    a, b = span(r)
    if (b - a) / 1e6 < args.ms or r["Function"] in skip:
        continue
    during_c = [c for c in copies if span(c)[1] > a and span(c)[0] < b]
    during_k = [k for k in kernels if span(k)[1] > a and span(k)[0] < b]
    busy = sorted([span(x) for x in during_c + during_k])
    covered, end = 0, a
    for s, e in busy:
        s, e = max(s, end), min(e, b)
        if e > s:
            covered += e - s
            end = e
    report["long_calls"].append({
        "function": r["Function"], "thread": r.get("Thread_Id"), "at_ms": (a - t0) / 1e6, "ms": (b - a) / 1e6,
        "device_busy_ms_during_call": covered / 1e6,
        "copies": [{"dir": c.get("Direction"), "start_after_call_ms": (span(c)[0] - a) / 1e6, "ms": (span(c)[1] - span(c)[0]) / 1e6} for c in during_c][:12],
        "kernels": [{"name": k.get("Kernel_Name", "")[:60], "start_after_call_ms": (span(k)[0] - a) / 1e6, "ms": (span(k)[1] - span(k)[0]) / 1e6} for k in during_k][:12],
        "api_calls_of_other_threads_meanwhile": sorted({o["Function"] for o in api if o is not r and o.get("Thread_Id") != r.get("Thread_Id") and span(o)[1] > a and span(o)[0] < b})[:20],
    })
print(json.dumps(report, indent=1))
