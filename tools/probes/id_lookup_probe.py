"""How long the distinct-id look-up of a fresh ranking takes on this host: a table of 8.84 M names, 8.67 M looked up in random
order (the cache flushed before each), then in table order; the cores the library believes it may use."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent / 'fast-forward-indexes_amd'))
import numpy as np
from fast_forward_amd import native
n=8_841_823
blob=("\n".join(f"D{i}" for i in range(n))).encode()
t=native.IdStrings()
t0=time.perf_counter(); t.intern_blob(blob,n); print("intern",time.perf_counter()-t0)
rng=np.random.default_rng(0)
flush=np.zeros(200_000_000,dtype=np.float32)
for rep in range(4):
    pick=rng.permutation(n)[:8_669_500]
    q=("\n".join(f"D{i}" for i in pick)).encode()
    flush+=1.0  # 800 MB read+write: the table leaves the cache
    t0=time.perf_counter(); out=t.lookup(range(len(pick)),missing_ok=True,packed=(q,len(pick))); dt=time.perf_counter()-t0
    print("lookup",round(dt,3), (out==pick).all())
seq=("\n".join(f"D{i}" for i in range(8_669_500))).encode()
for rep in range(2):
    flush+=1.0
    t0=time.perf_counter(); out=t.lookup(range(8_669_500),missing_ok=True,packed=(seq,8_669_500)); print("lookup in table order",round(time.perf_counter()-t0,3))
print("usable cores", native.usable_cores())
import os
print("cpu.max", open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else None, "affinity", len(os.sched_getaffinity(0)))
