import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import memb_amd
from memb_amd import synthetic
n=2196017
path,_=synthetic.cached_model(n,300,'trained',4)
out=torch.empty((n,300),dtype=torch.float32,device='cuda')
src=torch.randn((n,300),dtype=torch.float32,device='cuda')
rows=torch.arange(n,dtype=torch.int32,device='cuda')
def timeit(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a,b in ev:
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    ms=sorted(a.elapsed_time(b) for a,b in ev); return ms[0], ms[len(ms)//2]
print('fill_   (2.64 GB write)      min %.3f med %.3f ms' % timeit(lambda: out.fill_(1.0)))
print('copy_   (2.64 GB rd + wr)    min %.3f med %.3f ms' % timeit(lambda: out.copy_(src)))
for flags,name in ((0,'full'),(1,'no decode'),(2,'no output'),(3,'neither')):
    os.environ['MEMB_HIP_DEBUG']=str(flags)
    r=memb_amd.Reader(path,device=0)
    print('decode_trained %-10s     min %.3f med %.3f ms' % ((name,)+timeit(lambda: r.rows_embedding_device(rows,out=out))))
    del r
