#!/bin/bash
# VERDICT r03 #9, the one locality experiment: candidate lists in first-stage order vs sorted by row inside every query (=
# bucketed by page / by stack), on the 27 GB and on a 261 GB index, with the address-translation counters of the kernel
# (each in its own rocprofv3 --pmc pass).  Output: <out>/summary.json
set -u
OUT=${1:-gpurun_out/r04/locality}
BIG=${2:-85000000}
ROOT=$(pwd)
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
COUNTERS=$( (rocprofv3 --list-avail 2>/dev/null || rocprofv3 -L 2>/dev/null) | grep -o -E "Counter_Name +:[[:space:]]+(TCP_UTCL1_[A-Z_]+_sum|TCP_UTCL1_[A-Z_]+|TCC_HIT_sum|TCC_MISS_sum)" | awk '{print $NF}' | sort -u | grep -E "_sum$" | grep -E "TRANSLATION_MISS|TRANSLATION_HIT|UTCL1_REQUEST|PERMISSION_MISS|TCC_MISS" | head -5 | tr '\n' ' ')
echo "counters: $COUNTERS" | tee "$OUT/counters.txt"
for rows in 8841823 $BIG; do
  for order in first sorted in_order; do
    tag=${rows}_$order
    python3 $ROOT/tools/locality_probe.py --rows $rows --order $order > "$OUT/time_$tag.json" 2> "$OUT/time_$tag.err" || echo "FAILED time $tag"
    cat "$OUT/time_$tag.json"
    for c in $COUNTERS; do
      rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_${tag}_$c" -o pmc -- python3 $ROOT/tools/locality_probe.py --rows $rows --order $order --steps 2 > /dev/null 2> "$OUT/pmc_${tag}_$c.err" || echo "FAILED pmc $tag $c"
    done
  done
done
python3 - "$OUT" <<'PY'
import csv, json, sys
from pathlib import Path
out = Path(sys.argv[1]); res = {}
for f in sorted(out.glob("time_*.json")):
    try: res[f.stem[5:]] = json.loads(f.read_text().strip().splitlines()[-1])
    except Exception as e: res[f.stem[5:]] = {"error": str(e)}
for d in sorted(p for p in out.iterdir() if p.is_dir() and p.name.startswith("pmc_")):
    _, rows, rest = d.name.split("_", 2)
    order = "in_order" if rest.startswith("in_order") else rest.split("_", 1)[0]
    counter = rest[len(order) + 1:]
    per = {}
    for f in d.rglob("*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "gather_dot" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                per[(f.name, r["Dispatch_Id"])] = per.get((f.name, r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    if per: res.setdefault(f"{rows}_{order}", {}).setdefault("counters_per_launch", {})[counter] = sum(per.values()) / len(per)
for k, v in res.items():
    c = v.get("counters_per_launch", {})
    if "pairs" in v:
        v["counters_per_row"] = {n: x / v["pairs"] for n, x in c.items()}
json.dump(res, open(out / "summary.json", "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters_per_launch"} for k, v in res.items()}, indent=1))
PY
rm -rf "$OUT"/pmc_*/  # the raw csv files are large: the summary keeps the per-launch values
