import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import memb_amd, oracle
from memb_amd import synthetic
n=2196017
path,_=synthetic.cached_model(n,300,'trained',4)
t=time.time(); r=memb_amd.Reader(path); keys=r.keys(); print('open+keys %.2fs'%(time.time()-t))
t=time.time(); r.info(); print('stage to HBM %.2fs'%(time.time()-t))
rng=np.random.default_rng(3)
for m in (1, 1000, 100000, n):
    words = keys if m==n else [keys[i] for i in rng.integers(0,n,size=m)]
    best=[1e9]*3
    for rep in range(3):
        t0=time.perf_counter(); rows=r.resolve_rows(words); t1=time.perf_counter(); out=r.rows_embedding(rows); t2=time.perf_counter()
        del out   # (freeing a 2.6 GB result costs tens of ms; keep it out of the timed regions)
        t3=time.perf_counter(); full=r.batch_embedding(words); t4=time.perf_counter()
        del full
        best=[min(a,b) for a,b in zip(best,(t1-t0,t2-t1,t4-t3))]
    t0,t1,t2,t3=0,best[0],best[0]+best[1],best[0]+best[1]+best[2]
    print('n=%8d resolve %.4fs (%.2f Mw/s) | rows->numpy %.4fs (%.2f GB/s) | reader[words] %.4fs (%.3f M emb/s)'%(m,t1-t0,m/(t1-t0)/1e6,t2-t1,m*1200/(t2-t1)/1e9,t3-t2,m/(t3-t2)/1e6))
o=oracle.OracleReader(path, os.cpu_count())
if oracle.reference_available():
    ref=oracle.ReferenceDecoder(o)
    for m in (100000, n):
        rows=rng.integers(0,n,size=m).astype(np.uint32) if m<n else np.arange(n,dtype=np.uint32)
        out=np.empty((m,300),dtype=np.float32)
        for threads in (os.cpu_count(), 64, 16, 1):
            if threads==1 and m==n: continue
            best=1e9
            for rep in range(3):
                t=time.perf_counter(); ref.rows_embedding(rows,out=out,num_threads=threads); best=min(best,time.perf_counter()-t)
            print('reference decoder (oracle/_ref), decode only, n=%d threads=%d: %.4fs (%.1f M emb/s)'%(m,threads,best,m/best/1e6), flush=True)
        del out
words=[keys[i] for i in rng.integers(0,n,size=100000)]
t=time.time(); o.batch_embedding(words); print('oracle (all cores) batch 100k: %.4fs'%(time.time()-t))
o1=oracle.OracleReader(path, 1)
t=time.time(); o1.batch_embedding(words[:20000]); print('oracle (1 thread) batch 20k: %.4fs'%(time.time()-t))
