"""Persistent vs one-shot decode kernel over batch sizes (random rows, device resident)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import memb_amd
from memb_amd import synthetic
n=2196017
path,_=synthetic.cached_model(n,300,'trained',4)
rng=np.random.default_rng(11)
def timeit(f, reps=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a,b in ev:
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    ms=sorted(a.elapsed_time(b) for a,b in ev); return ms[0], ms[len(ms)//2]
readers={}
for pers in (1,0):
    os.environ['MEMB_HIP_PERSISTENT']=str(pers)
    readers[pers]=memb_amd.Reader(path,device=0); readers[pers].info()
for m in (1000, 10000, 50000, 100000, 300000, 1000000):
    rows=torch.from_numpy(rng.integers(0,n,size=m).astype(np.int32)).cuda()
    out=torch.empty((m,300),device='cuda')
    line='n=%8d'%m
    for pers in (1,0):
        os.environ['MEMB_HIP_PERSISTENT']=str(pers)
        r=readers[pers]
        mn,med=timeit(lambda: r.rows_embedding_device(rows,out=out))
        line+=' | %s min %.4f med %.4f ms (%.2f G emb/s)'%('persistent' if pers else 'one-shot  ',mn,med,m/med/1e6)
    print(line,flush=True)
