#!/bin/bash
# rocprofv3 evidence for profiles/r02 (run on the GPU box from the repo root):
#   1. --kernel-trace --stats of `bench.py --no-regions` (kernel time must agree with the bench line's HIP events)
#   2. --pmc passes (own runs, never combined with tracing): FETCH_SIZE, WRITE_SIZE, TCC hit/miss/requests
#      for the bench, for the any-d kernel (d = 100 / 300 / 1000) and for PQ decode-then-dot.
set -u
OUT=${1:-gpurun_out/r02_prof}
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
DIMS="python3 $ROOT/tools/dims_bench.py 100,300,1000"
run dims_fetch --pmc FETCH_SIZE --output-format csv -d "$OUT/dims_fetch" -o pmc -- $DIMS
run dims_tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$OUT/dims_tcc" -o pmc -- $DIMS
PQ="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-regions --no-traffic --cpu-sample-queries 0 --store pq"
export FFX_PQ_NO_ADC=1
run pq_decode_fetch --pmc FETCH_SIZE --output-format csv -d "$OUT/pq_decode_fetch" -o pmc -- $PQ
run pq_decode_tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$OUT/pq_decode_tcc" -o pmc -- $PQ
run pq_decode_lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE SQ_INSTS_LDS --output-format csv -d "$OUT/pq_decode_lds" -o pmc -- $PQ
unset FFX_PQ_NO_ADC
find "$OUT" -name "*.csv" | head -40
