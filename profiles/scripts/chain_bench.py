import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from leon_amd import capi
if os.environ.get('LEON_LIB'):
    capi.lib_path = lambda: os.environ['LEON_LIB']
capi.load_library()
rng=np.random.default_rng(1)
n=int(sys.argv[1]) if len(sys.argv)>1 else 4_000_000
k=31
km=rng.integers(0,1<<62,n,dtype=np.uint64)
best=1e9
for _ in range(3):
    t=time.perf_counter(); out=capi.host_anchor_dict_encode(km,k); dt=time.perf_counter()-t; best=min(best,dt)
import hashlib
print("symbols %d  best %.3f s  %.2f ns/symbol  sha %s"%(n*k,best,best/(n*k)*1e9,hashlib.sha256(out).hexdigest()[:16]))
