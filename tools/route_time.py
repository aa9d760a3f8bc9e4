import sys, time, numpy as np
sys.path.insert(0,'/root/repo/fast-forward-indexes_amd'); sys.path.insert(0,'/root/repo')
from fast_forward_amd import native
import bench as B
rows,nq,k=8_841_823,6980,5000
ids,sparse=B.candidates(rows,nq,k,2024)
pair_off=np.arange(nq+1,dtype=np.int64)*k
for world in (1,2,8):
    bounds=(np.arange(world+1,dtype=np.int64)*rows)//world
    for _ in range(2):
        t=time.perf_counter(); r=native.route(pair_off,ids,0,world,doc_bounds=bounds); dt=time.perf_counter()-t
    print(world, round(dt,4), {k_:(v.shape if hasattr(v,'shape') else v) for k_,v in r.items()} if world==8 else "")
