"""Kernel-time survey over BASELINE.json's configurations (device-resident rows/output, HIP events)."""
import os, sys, time, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import memb_amd
from memb_amd import synthetic
lib = ctypes.CDLL(memb_amd.HIP_LIBRARY_PATH)
def algbytes(reader, rows_host):
    b = ctypes.c_uint64(0)
    lib.memb_hip_algorithmic_bytes(ctypes.c_void_p(reader._impl.context_handle()), rows_host.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(len(rows_host)), ctypes.byref(b))
    return b.value
def timeit(f, reps=15):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a,b in ev:
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    ms=sorted(a.elapsed_time(b) for a,b in ev); return ms[len(ms)//2]
def report(name, reader, rows_host, out, col_off=0):
    rows = torch.from_numpy(rows_host.view(np.int32)).cuda()
    ms = timeit(lambda: reader.rows_embedding_device(rows, out=out, col_off=col_off))
    ab = algbytes(reader, rows_host)
    print('%-44s n=%8d  %.3f ms  %6.2f G emb/s  %5.2f TB/s alg (%4.1f%% of 8 TB/s)  %s' % (name, len(rows_host), ms, len(rows_host)/ms/1e6, ab/ms/1e9, ab/ms/1e9/8000*100, {k:reader.info()[k] for k in ('lanes_per_word','segment_symbols','waves_per_block','max_code_bits','root_bits','max_stream_bytes')}), flush=True)
which = os.environ.get('CFG', 'h,c2,c3,c4,c5,b8,u8').split(',')
N1, N2 = 2196017, 1999995
rng = np.random.default_rng(11)
if 'h' in which or 'c2' in which or 'c5' in which:
    p4,_ = synthetic.cached_model(N1, 300, 'trained', 4); r4 = memb_amd.Reader(p4, device=0)
if 'h' in which:
    report('H  glove 4-bit full dump', r4, np.arange(N1, dtype=np.uint32), torch.empty((N1,300), device='cuda'))
if 'c2' in which:
    rows = rng.integers(0, N1, size=100000).astype(np.uint32); rows[rng.integers(0,100000,size=1000)] = 0xFFFFFFFF
    report('C2 glove 4-bit 100k random batch (1% miss)', r4, rows, torch.empty((100000,300), device='cuda'))
    rows = rng.integers(0, N1, size=1000).astype(np.uint32)
    report('   glove 4-bit 1k random batch', r4, rows, torch.empty((1000,300), device='cuda'))
if 'c3' in which:
    p,_ = synthetic.cached_model(N2, 300, 'trained', 6); r = memb_amd.Reader(p, device=0)
    report('C3 fasttext-shape 6-bit full dump', r, np.arange(N2, dtype=np.uint32), torch.empty((N2,300), device='cuda'))
if 'c4' in which:
    p,_ = synthetic.cached_model(N1, 300, 'trained', 2); r = memb_amd.Reader(p, device=0)
    report('C4 glove 2-bit full dump (1 GPU)', r, np.arange(N1, dtype=np.uint32), torch.empty((N1,300), device='cuda'))
if 'c5' in which:
    pb,_ = synthetic.cached_model(N2, 300, 'trained', 4, seed=77); rb = memb_amd.Reader(pb, device=0)
    n = 500000
    out = torch.empty((n,600), device='cuda')
    rows_a = rng.integers(0, N1, size=n).astype(np.uint32); rows_a[rng.random(n) < 0.25] = 0xFFFFFFFF
    rows_b = rng.integers(0, N2, size=n).astype(np.uint32); rows_b[rng.random(n) < 0.25] = 0xFFFFFFFF
    report('C5 concat half A (ld 600, col 0)', r4, rows_a, out, 0)
    report('C5 concat half B (ld 600, col 300)', rb, rows_b, out, 300)
if 'b8' in which:
    p,_ = synthetic.cached_model(N1, 300, 'trained', 8); r = memb_amd.Reader(p, device=0)
    report('   glove 8-bit full dump', r, np.arange(N1, dtype=np.uint32), torch.empty((N1,300), device='cuda'))
if 'u8' in which:
    nu = 500000
    p,_ = synthetic.cached_model(nu, 300, 'uniform', 8); r = memb_amd.Reader(p, device=0)
    report('   uniform 8-bit full dump (500k)', r, np.arange(nu, dtype=np.uint32), torch.empty((nu,300), device='cuda'))
    p,_ = synthetic.cached_model(nu, 300, 'full', 8); r = memb_amd.Reader(p, device=0)
    report('   full fp32 dump (500k)', r, np.arange(nu, dtype=np.uint32), torch.empty((nu,300), device='cuda'))
