#!/bin/bash
# rocprofv3 evidence for profiles/r03 (run on the GPU box from the repo root):
#   1. --kernel-trace --stats of `bench.py --no-regions` (kernel time must agree with the bench line's HIP events)
#   2. --pmc passes (own runs, never combined with tracing): FETCH_SIZE, WRITE_SIZE, TCC hit/miss/requests for the bench
#   3. --pmc FETCH_SIZE (+ TCC) per row length for the kernels of this round: long rows through segments
#      (d = 1100 / 2000 / 5000) and the short rows whose lines straddle (d = 100 / 200 / 300)
set -u
OUT=${1:-gpurun_out/r03_prof}
ROOT=$(pwd)
mkdir -p "$OUT"
OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
run() { # name, rocprof args..., -- , command...
  local name=$1; shift
  rocprofv3 "$@" > "$OUT/$name.stdout" 2> "$OUT/$name.stderr" || echo "FAILED: $name" | tee -a "$OUT/failures.txt"
  echo "done: $name"
}
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 5 --no-regions --no-traffic --cpu-sample-queries 0"
run bench_trace --kernel-trace --stats --output-format csv -d "$OUT/bench_trace" -o bench -- $BENCH
SHORT="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-regions --no-traffic --cpu-sample-queries 0"
run bench_fetch --pmc FETCH_SIZE --output-format csv -d "$OUT/bench_fetch" -o pmc -- $SHORT
run bench_write --pmc WRITE_SIZE --output-format csv -d "$OUT/bench_write" -o pmc -- $SHORT
run bench_tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$OUT/bench_tcc" -o pmc -- $SHORT
for d in 1100 2000 5000 100 200 300; do
  DIMS="python3 $ROOT/tools/dims_bench.py $d --only-shipping"
  run dims${d}_fetch --pmc FETCH_SIZE --output-format csv -d "$OUT/dims${d}_fetch" -o pmc -- $DIMS
  run dims${d}_tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$OUT/dims${d}_tcc" -o pmc -- $DIMS
done
find "$OUT" -name "*.csv" | wc -l
