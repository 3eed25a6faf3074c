"""uniform / full kernels: dump and random rows, kernel time (steady state). Run once per library build."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import memb_amd
from memb_amd import synthetic
n = 500000
for storage in ('uniform', 'full'):
    path, _ = synthetic.cached_model(n, 300, storage, 8)
    r = memb_amd.Reader(path, device=0); r.info()
    out = torch.empty((n, 300), dtype=torch.float32, device='cuda')
    for name, rows in (('dump', torch.arange(n, dtype=torch.int32, device='cuda')), ('random', torch.randperm(n, device='cuda').to(torch.int32))):
        f = lambda: r.rows_embedding_device(rows, out=out)
        for _ in range(150): f()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
        for a, b in ev:
            a.record(); f(); b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev)
        print('%-8s %-7s median %.4f ms min %.4f' % (storage, name, ms[15], ms[0]), flush=True)
