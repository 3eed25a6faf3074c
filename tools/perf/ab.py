import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import memb_amd
from memb_amd import synthetic
n=2196017
bits=int(os.environ.get('AB_BITS','4'))
path,_=synthetic.cached_model(n,300,'trained',bits)
out=torch.empty((n,300),dtype=torch.float32,device='cuda')
rows=torch.arange(n,dtype=torch.int32,device='cuda')
perm=torch.randperm(n,device='cuda').to(torch.int32)
small=perm[:100000].contiguous(); small_out=torch.empty((100000,300),dtype=torch.float32,device='cuda')
def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a,b in ev:
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    ms=sorted(a.elapsed_time(b) for a,b in ev); return ms[0], ms[len(ms)//2]
variants=[tuple(v.split(':')) for v in os.environ.get('AB','nt:0,plain:4').split(',')]
readers={}
for name,flags in variants:
    os.environ['MEMB_HIP_DEBUG']=flags
    for kv in os.environ.get('AB_ENV_'+name,'').split():
        k,v=kv.split('='); os.environ[k]=v
    readers[name]=memb_amd.Reader(path,device=0); readers[name].info()
for rnd in range(3):
    for name,flags in variants:
        os.environ['MEMB_HIP_DEBUG']=flags
        r=readers[name]
        a=timeit(lambda: r.rows_embedding_device(rows,out=out)); b=timeit(lambda: r.rows_embedding_device(perm,out=out)); c=timeit(lambda: r.rows_embedding_device(small,out=small_out))
        print('round %d %-8s sorted min %.3f med %.3f | random min %.3f med %.3f | 100k random min %.4f med %.4f ms' % ((rnd,name)+a+b+c), flush=True)
