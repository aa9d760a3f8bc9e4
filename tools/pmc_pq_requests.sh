#!/bin/bash
# What a product-quantized row costs the fabric: read requests of the fused exact-ADC kernel, each counter in
# its own rocprofv3 --pmc pass (never combined with tracing), per launch:
#   TCC_EA0_RDREQ_sum      all read requests L2 -> memory     TCC_EA0_RDREQ_32B_sum  the 32-byte ones     TCC_BUBBLE_sum  the 128-byte ones
# (FETCH_SIZE = (BUBBLE*128 + (RDREQ - BUBBLE - RDREQ_32B)*64 + RDREQ_32B*32) / 1024 is the tool's own expression.)
set -u
OUT=${1:-gpurun_out/r04/pmc_pq}
ROOT=$(pwd)
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
for store in pq f32; do
  CMD="python3 $ROOT/bench.py --store $store --steps 3 --warmup 1 --no-regions --no-traffic --cpu-sample-queries 0"
  for c in TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum FETCH_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d "$OUT/${store}_$c" -o pmc -- $CMD > "$OUT/${store}_$c.stdout" 2> "$OUT/${store}_$c.stderr" || echo "FAILED $store $c"
  done
done
python3 - "$OUT" <<'PY'
import csv, json, sys
from pathlib import Path
out = Path(sys.argv[1]); res = {}
for d in sorted(p for p in out.iterdir() if p.is_dir()):
    store, _, counter = d.name.partition("_")
    want = "pq_adc" if store == "pq" else "gather_dot"
    per = {}
    for f in d.rglob("*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if want in r["Kernel_Name"] and r["Counter_Name"] == counter:
                per[(f.name, r["Dispatch_Id"])] = per.get((f.name, r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    if per:
        res.setdefault(store, {})[counter] = sum(per.values()) / len(per)
rows = 6980 * 5000
for store, c in res.items():
    c["per_row"] = {k: v / rows for k, v in c.items() if k != "FETCH_SIZE"}
    if "FETCH_SIZE" in c: c["FETCH_SIZE_bytes_per_row_x1"] = c["FETCH_SIZE"] * 1024 / rows
print(json.dumps(res, indent=1))
json.dump(res, open(out / "summary.json", "w"), indent=1)
PY
