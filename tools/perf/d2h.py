"""rows -> numpy through the host-buffer entry point (MEMB_HIP_VERBOSE=1 prints the library's phases)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import memb_amd
from memb_amd import synthetic
n=2196017
path,_=synthetic.cached_model(n,300,'trained',4)
r=memb_amd.Reader(path); r.info()
rows=np.arange(n,dtype=np.uint32)
for m in (100000, n, n):
    t=time.time(); out=r.rows_embedding(rows[:m]); dt=time.time()-t
    print('rows->numpy n=%d %.4fs %.1f GB/s'%(m,dt,m*1200/dt/1e9), flush=True)
    del out
