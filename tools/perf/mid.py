"""Mid-size batches through the host-buffer entry point: where do the milliseconds go?"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import ctypes
import numpy as np
import memb_amd
from memb_amd import synthetic
n=2196017
path,_=synthetic.cached_model(n,300,'trained',4)
r=memb_amd.Reader(path); r.info()
lib=ctypes.CDLL(memb_amd.HIP_LIBRARY_PATH)
lib.memb_hip_decode_rows.argtypes=[ctypes.c_void_p,ctypes.c_void_p,ctypes.c_size_t,ctypes.c_void_p,ctypes.c_size_t,ctypes.c_size_t]
handle=r._impl.context_handle()
rng=np.random.default_rng(1)
print(open('/sys/kernel/mm/transparent_hugepage/enabled').read().strip(), '| cpus', os.cpu_count())
for m in (10000, 100000, 400000, n):
    rows=rng.integers(0,n,size=m).astype(np.uint32) if m<n else np.arange(n,dtype=np.uint32)
    r.rows_embedding(rows)
    for threads in (0, 4, 8, 16, 32, 64):
        os.environ['MEMB_HIP_COPY_THREADS']=str(threads)
        r=memb_amd.Reader(path); r.info(); handle=r._impl.context_handle()   # switches are read when a context is created
        r.rows_embedding(rows)
        best=1e9; reuse=1e9
        out=np.empty((m,300),dtype=np.float32)
        for rep in range(5):
            t=time.perf_counter(); res=r.rows_embedding(rows); best=min(best,time.perf_counter()-t); del res
            t=time.perf_counter(); lib.memb_hip_decode_rows(handle, rows.ctypes.data, m, out.ctypes.data, 300, 0); reuse=min(reuse,time.perf_counter()-t)
        print('n=%7d threads=%2d fresh result %.2f ms (%.1f GB/s) | reused buffer %.2f ms'%(m,threads,best*1e3,m*1200/best/1e9,reuse*1e3), flush=True)
t=time.perf_counter(); a=np.empty((100000,300),dtype=np.float32); a[:]=1; print('numpy alloc+fill 120 MB: %.2f ms'%((time.perf_counter()-t)*1e3))
