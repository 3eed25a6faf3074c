import os, sys
sys.path.insert(0, os.getcwd())
os.environ['MEMB_HIP_VERBOSE'] = '1'
import numpy as np, torch
import memb_amd
from memb_amd import synthetic
for bits in (2, 6):
    n = 2196017 if bits == 2 else 1999995
    path, _ = synthetic.cached_model(n, 300, 'trained', bits)
    reader = memb_amd.Reader(path, device=0)
    rows = torch.arange(n, dtype=torch.int32, device='cuda')
    out = torch.empty((n, 300), dtype=torch.float32, device='cuda')
    print(bits, 'before', reader.info(n)['kernel'], flush=True)
    reader.rows_embedding_device(rows, out=out); torch.cuda.synchronize()
    print(bits, 'after ', reader.info(n)['kernel'], reader.info()['kernel'], flush=True)
    def t():
        for _ in range(40): reader.rows_embedding_device(rows, out=out)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): reader.rows_embedding_device(rows, out=out)
        b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / 20
    print(bits, 'auto', t())
    for forced in (2, 0, 1):
        reader.set_option('persistent', forced); print(bits, 'persistent option', forced, reader.info(n)['kernel'], t(), flush=True)
