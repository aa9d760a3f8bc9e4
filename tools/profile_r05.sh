#!/bin/bash
# The measurements behind MEASUREMENTS.md's round-5 numbers (run on the GPU box from the repo root); everything lands in <out>,
# the files that are kept are copied to profiles/r05/ by hand (profiles/r05/README.md names the command behind each).
#   tools/profile_r05.sh <out> [part ...]     parts: bench trace rehearsal sharded1 stores capacity ondisk
set -u
# (a part that fails ends the script: no further GPU step after a failed one)
OUT=${1:-gpurun_out/r05/final}; shift || true
PARTS=${*:-bench trace rehearsal sharded1 stores}
ROOT=$(pwd)
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
has() { [[ " $PARTS " == *" $1 "* ]]; }
say() { echo "== $* ($(date +%H:%M:%S))"; }
if has bench; then
  say "bench, N = 1, default (regions + PMC traffic)"
  python3 bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err" || { echo "FAILED bench_n1"; exit 1; }
fi
if has trace; then
  say "rocprofv3 --kernel-trace --stats of bench --no-regions"
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_trace" -o bench -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-regions --no-traffic --cpu-sample-queries 0 > "$OUT/bench_traced.json" 2> "$OUT/bench_traced.err") || { echo "FAILED trace"; exit 1; }
  find "$OUT/bench_trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_kernel_stats.csv" \;
  rm -rf "$OUT/bench_trace"
fi
if has rehearsal; then
  say "rehearsal of one rank of N = 1, 2, 4, 8 (copy stand-in)"
  python3 tools/rehearse_scaling.py --ranks 1,2,4,8 --steps 60 --warmup 6 --out "$OUT/rehearsal_scaling.json" > "$OUT/rehearsal_scaling.txt" 2> "$OUT/rehearsal_scaling.err" || { echo "FAILED rehearsal"; exit 1; }
  say "... paced stand-in (persistent workgroups spinning on flags, 50 GB/s per link)"
  python3 tools/rehearse_scaling.py --ranks 2,4,8 --steps 60 --warmup 6 --out "$OUT/rehearsal_scaling_paced.json" -- --rehearsal-exchange paced:50 > "$OUT/rehearsal_scaling_paced.txt" 2> "$OUT/rehearsal_scaling_paced.err" || { echo "FAILED rehearsal paced"; exit 1; }
  say "... query vectors replicated by H2D instead of all-gathered, N = 8"
  python3 tools/rehearse_scaling.py --ranks 8 --steps 60 --warmup 6 --out "$OUT/rehearsal_n8_replicated_queries.json" -- --fresh-queries replicate > "$OUT/rehearsal_n8_replicated_queries.txt" 2>&1 || { echo "FAILED rehearsal replicate"; exit 1; }
  say "... last rank of 8"
  python3 tools/rehearse_scaling.py --ranks 8 --which last --steps 60 --warmup 6 --out "$OUT/rehearsal_n8_last_rank.json" > "$OUT/rehearsal_n8_last_rank.txt" 2>&1 || { echo "FAILED rehearsal last"; exit 1; }
  say "... kernel + copy trace of the N = 8 rehearsal"
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/reh8_trace" -o t -- python3 $ROOT/bench.py --as-rank-of 8 --steps 12 --warmup 3 --cpu-sample-queries 0 > /dev/null 2> "$OUT/reh8_trace.err") || { echo "FAILED reh8 trace"; exit 1; }
  python3 - "$OUT" <<'PY'
import csv, sys
from pathlib import Path
out = Path(sys.argv[1]); d = out / "reh8_trace"
K = [r for f in d.rglob("*kernel_trace.csv") for r in csv.DictReader(open(f))]
M = [r for f in d.rglob("*memory_copy_trace.csv") for r in csv.DictReader(open(f))]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "kernel", r["Kernel_Name"][:70], r.get("Queue_Id", "")) for r in K]
ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", r.get("Direction", ""), "") for r in M]
ev.sort()
g = [i for i, e in enumerate(ev) if "gather_dot" in e[3]]
# three consecutive steps out of the fresh-batch loop (the gather_dots with routing kernels between them)
own = [i for i, e in enumerate(ev) if "own_place_kernel" in e[3]]
if own and len(own) > 6:
    mid = own[len(own) // 2]
    lo = max(j for j in g if j < mid)
    hi = [j for j in g if j > mid][2] if len([j for j in g if j > mid]) > 2 else g[-1]
    t0 = ev[lo][0]
    with open(out / "rehearsal_trace_n8_fresh_batches.csv", "w") as f:
        f.write("start_us,duration_us,kind,queue,name\n")
        for e in ev[lo:hi + 1]:
            f.write(f"{(e[0] - t0) / 1e3:.1f},{(e[1] - e[0]) / 1e3:.1f},{e[2]},{e[4]},\"{e[3]}\"\n")
PY
  rm -rf "$OUT/reh8_trace"
fi
if has sharded1; then
  say "the sharded leg with one rank over RCCL (--force-sharded): fresh batches and host-to-host through ffx_sharded_score"
  python3 bench.py --force-sharded --steps 30 --warmup 4 --cpu-sample-queries 16 --cpu-seconds 1 > "$OUT/bench_sharded_1rank.json" 2> "$OUT/bench_sharded_1rank.err" || { echo "FAILED sharded1"; exit 1; }
fi
if has stores; then
  for spec in "f16:--store f16" "pq:--store pq" "k1000:--k 1000" "multi:--layout multi" "f16_50m:--store f16 --rows 50000000" "pq_50m:--store pq --rows 50000000"; do
    name=${spec%%:*}; flags=${spec#*:}
    say "bench $flags"
    python3 bench.py $flags --no-regions --steps 30 --warmup 5 --cpu-sample-queries 32 --cpu-seconds 2 > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err" || { echo "FAILED bench_$name"; exit 1; }
  done
fi
if has capacity; then
  say "85 M x 768 fp32 = 261 GB resident"
  python3 bench.py --rows 85000000 --no-regions --no-traffic --steps 20 --warmup 3 --cpu-sample-queries 16 --cpu-seconds 1 > "$OUT/bench_85m_rows_261GB.json" 2> "$OUT/bench_85m.err" || { echo "FAILED capacity"; exit 1; }
fi
if has ondisk; then
  say "streamed index: tools/ondisk_bench.py"
  python3 tools/ondisk_bench.py --dir /dev/shm > "$OUT/ondisk_stream.json" 2> "$OUT/ondisk_stream.err" || { echo "FAILED ondisk"; exit 1; }
fi
say done
ls -la "$OUT" | head -60
