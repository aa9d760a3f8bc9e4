#!/usr/bin/env python3
"""gather_dot at row lengths that have no kernel of their own (the "any d" strided kernel, and the staged
generic kernel it replaces: FFX_DEBUG_VARIANT=generic_staged), index ~13 GB each, 2000 queries x k=5000."""
import json
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "fast-forward-indexes_amd"))
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import _experiments  # noqa: E402

_experiments.use()  # the switches this tool flips exist in the experiments build only
from fast_forward_amd import native, synth  # noqa: E402
import bench as B  # noqa: E402

out = {}
ONLY_SHIPPING = "--only-shipping" in sys.argv  # (PMC passes: just the kernel that ships, 3 launches)
for d in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "32,100,200,300,1000,768").split(",")]:
    n = min(40_000_000, int(13e9 / (d * 4)))
    nq, k = 2000, 5000
    idx = native.DeviceIndex(n, d)
    idx.fill_synthetic(1, 0.05, 0, n)
    idx.set_doc_rows(np.arange(n + 1, dtype=np.int64), np.arange(n, dtype=np.int32))
    idx.set_psg_rows(np.arange(n, dtype=np.int32))
    ids, sparse = B.candidates(n, nq, k, 3)
    plan = native.Plan(idx)
    plan.set_queries(synth.rows(2, 0.05, 0, nq, d))
    plan.set_pairs(np.arange(nq + 1, dtype=np.int64) * k, ids, sparse, 1)
    res = {}
    for label, env in ((("anyd", None),) if ONLY_SHIPPING else (("anyd", None), ("anyd_u2", "u2"), ("anyd_u8", "u8"), ("staged", "1"))):
        os.environ.pop("FFX_DEBUG_VARIANT", None)
        if env == "1":
            os.environ["FFX_DEBUG_VARIANT"] = "generic_staged"
        elif env:
            os.environ["FFX_DEBUG_VARIANT"] = f"anyd_u={env[1:]}"
        ms = []
        for _ in range(3 if ONLY_SHIPPING else 6):
            plan.run(0.1, exact_sum=True)
            ms.append(plan.stats()["ms_gather_dot"])
        st = plan.stats()
        ff = plan.fetch()[0]
        res[label] = {"ms": float(np.mean(ms[1:])), "GBs": nq * k * (d * 4 + 4) / float(np.mean(ms[1:])) / 1e6, "variant": st["kernel_variant"],
                      "checksum": float(ff.astype(np.float64).sum())}
    os.environ.pop("FFX_DEBUG_VARIANT", None)
    res["rows_per_launch"], res["algorithmic_bytes_per_launch"] = nq * k, nq * k * (d * 4 + 4)
    res["same_bits"] = ONLY_SHIPPING or res["anyd"]["checksum"] == res["staged"]["checksum"]
    out[str(d)] = res
    plan.close()
    idx.close()
print(json.dumps(out, indent=1))
