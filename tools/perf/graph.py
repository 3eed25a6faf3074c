"""Device entry point under HIP graph capture (torch.cuda.CUDAGraph): eager launches against replay."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import memb_amd
from memb_amd import synthetic
n=2196017
path,_=synthetic.cached_model(n,300,'trained',4)
r=memb_amd.Reader(path,device=0); r.info()
rng=np.random.default_rng(2)
for m in (64, 1000, 10000, 100000):
    rows=torch.from_numpy(rng.integers(0,n,size=m).astype(np.uint32).view(np.int32)).cuda()
    out=torch.empty((m,300),dtype=torch.float32,device='cuda')
    for _ in range(3): r.rows_embedding_device(rows,out=out)
    torch.cuda.synchronize(); ref=out.clone()
    def eager(k):
        for _ in range(k): r.rows_embedding_device(rows,out=out)
    side=torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3): r.rows_embedding_device(rows,out=out)
    torch.cuda.synchronize()
    graph=torch.cuda.CUDAGraph()
    out.zero_()
    with torch.cuda.graph(graph, stream=side):
        r.rows_embedding_device(rows,out=out)
    graph.replay(); torch.cuda.synchronize()
    same=bool(torch.equal(out.view(torch.int32),ref.view(torch.int32)))
    k=200
    torch.cuda.synchronize(); t=time.perf_counter(); eager(k); torch.cuda.synchronize(); te=(time.perf_counter()-t)/k
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(k): graph.replay()
    torch.cuda.synchronize(); tg=(time.perf_counter()-t)/k
    print('n=%6d eager %.1f us/call, graph replay %.1f us/call, replay gives the same bits: %s'%(m,te*1e6,tg*1e6,same),flush=True)
