"""Time decode_trained under several tile geometries (env overrides), one process."""
import os, sys, json, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import memb_amd
from memb_amd import synthetic
bits = int(os.environ.get('TUNE_BITS', '4'))
path, _ = synthetic.cached_model(2196017, 300, 'trained', bits)
configs = [tuple(map(int, c.split('x'))) for c in os.environ.get('TUNE', '4x300,2x300,1x300,4x152,4x100,2x100,4x76,4x60,4x44,2x44,4x32,1x32').split(',')]
n = 2196017
rows = torch.arange(n, dtype=torch.int32, device='cuda')
perm = torch.randperm(n, device='cuda').to(torch.int32)
out = torch.empty((n, 300), dtype=torch.float32, device='cuda')
results = []
for waves, chunk in configs:
    os.environ['MEMB_HIP_WAVES'] = str(waves); os.environ['MEMB_HIP_CHUNK'] = str(chunk)
    reader = memb_amd.Reader(path, device=0)
    info = reader.info()
    for name, r in (('sorted', rows), ('random', perm)):
        for _ in range(3): reader.rows_embedding_device(r, out=out)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a, b in ev:
            a.record(); reader.rows_embedding_device(r, out=out); b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev)
        results.append((waves, chunk, info['lds_bytes_per_block'], name, ms[0], ms[len(ms)//2]))
        print('waves %d chunk %3d lds %6d %-6s min %.3f ms med %.3f ms  -> %.2f TB/s' % (waves, chunk, info['lds_bytes_per_block'], name, ms[0], ms[len(ms)//2], 2.941/ms[len(ms)//2]), flush=True)
    del reader
