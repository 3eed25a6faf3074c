"""configs[4] (concatenation of two 4-bit models, 500 k rows, ld 600): two launches over the whole batch
against the two readers alternating over row chunks."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import memb_amd
from memb_amd import synthetic
pa,_=synthetic.cached_model(2196017,300,'trained',4)
pb,_=synthetic.cached_model(1999995,300,'trained',4,seed=4321)
a=memb_amd.Reader(pa,device=0); b=memb_amd.Reader(pb,device=0); a.info(); b.info()
n=500000
rng=np.random.default_rng(5)
ra=rng.integers(0,2196017,size=n).astype(np.uint32); ra[rng.random(n)<0.25]=0xFFFFFFFF
rb=rng.integers(0,1999995,size=n).astype(np.uint32); rb[rng.random(n)<0.25]=0xFFFFFFFF
ta=torch.from_numpy(ra.view(np.int32)).cuda(); tb=torch.from_numpy(rb.view(np.int32)).cuda()
out=torch.empty((n,600),dtype=torch.float32,device='cuda')
def whole():
    a.rows_embedding_device(ta,out=out,col_off=0); b.rows_embedding_device(tb,out=out,col_off=300)
def chunked(rows):
    def run():
        for s in range(0,n,rows):
            e=min(n,s+rows)
            a.rows_embedding_device(ta[s:e],out=out[s:e],col_off=0); b.rows_embedding_device(tb[s:e],out=out[s:e],col_off=300)
    return run
def timeit(f,reps=15):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for x,y in ev:
        x.record(); f(); y.record()
    torch.cuda.synchronize()
    ms=sorted(x.elapsed_time(y) for x,y in ev); return ms[len(ms)//2]
whole(); torch.cuda.synchronize(); ref=out.clone()
print('two launches over the whole batch: %.3f ms'%timeit(whole))
for rows in (250000,125000,65536,32768,16384):
    t=timeit(chunked(rows)); chunked(rows)(); torch.cuda.synchronize()
    print('alternating over chunks of %6d rows: %.3f ms  same bits: %s'%(rows,t,bool(torch.equal(out.view(torch.int32),ref.view(torch.int32)))))
dense=torch.empty((n,300),dtype=torch.float32,device='cuda')
print('one reader, dense (n, 300) output: %.3f ms'%timeit(lambda: a.rows_embedding_device(ta,out=dense)))
wide=torch.empty((n,608),dtype=torch.float32,device='cuda')
print('two launches, ld 608 (rows 32-B aligned): %.3f ms'%timeit(lambda: (a.rows_embedding_device(ta,out=wide,col_off=0), b.rows_embedding_device(tb,out=wide,col_off=304))))
wide=torch.empty((n,640),dtype=torch.float32,device='cuda')
print('two launches, ld 640, col_off 0 / 320 (halves 128-B aligned): %.3f ms'%timeit(lambda: (a.rows_embedding_device(ta,out=wide,col_off=0), b.rows_embedding_device(tb,out=wide,col_off=320))))
u=memb_amd.ReadersUnion([a,b],'concatenate')
from memb_amd import _memb
def fused():
    assert _memb.union_rows_to_device([a._impl,b._impl],[ta.data_ptr(),tb.data_ptr()],[0,300],n,out.data_ptr(),600,torch.cuda.current_stream().cuda_stream)
t=timeit(fused); out.zero_(); fused(); torch.cuda.synchronize()
print('ONE fused launch (decode_trained_union): %.3f ms  same bits: %s'%(t,bool(torch.equal(out.view(torch.int32),ref.view(torch.int32)))))
s1=torch.cuda.Stream(); s2=torch.cuda.Stream()
def concurrent():
    cur=torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): a.rows_embedding_device(ta,out=out,col_off=0)
    with torch.cuda.stream(s2): b.rows_embedding_device(tb,out=out,col_off=300)
    cur.wait_stream(s1); cur.wait_stream(s2)
t=timeit(concurrent); concurrent(); torch.cuda.synchronize()
print('two launches on two streams (concurrent): %.3f ms  same bits: %s'%(t,bool(torch.equal(out.view(torch.int32),ref.view(torch.int32)))))
for ld in (300, 304, 320, 400, 600, 1200):
    wide=torch.empty((n,ld),dtype=torch.float32,device='cuda')
    print('one reader, ld %4d: %.3f ms'%(ld,timeit(lambda: a.rows_embedding_device(ta,out=wide,col_off=0))))
    del wide
avg=torch.empty((n,300),dtype=torch.float32,device='cuda')
def average_two():
    a.rows_embedding_device(ta,out=avg); b.rows_embedding_device(tb,out=avg,accumulate=True,divisor=2.0)
def average_fused():
    assert _memb.union_rows_to_device([a._impl,b._impl],[ta.data_ptr(),tb.data_ptr()],[0,0],n,avg.data_ptr(),300,torch.cuda.current_stream().cuda_stream,True)
average_two(); torch.cuda.synchronize(); want=avg.clone()
t2=timeit(average_two); t1=timeit(average_fused); average_fused(); torch.cuda.synchronize()
print('average of the two models: two launches %.3f ms, one fused launch %.3f ms, same bits: %s'%(t2,t1,bool(torch.equal(avg.view(torch.int32),want.view(torch.int32)))))
