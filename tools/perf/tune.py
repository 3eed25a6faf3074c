"""Time decode_trained under several tile geometries (env overrides), one process."""
import os, sys, json, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import memb_amd
from memb_amd import synthetic
bits = int(os.environ.get('TUNE_BITS', '4'))
n = int(os.environ.get('TUNE_WORDS', '2196017'))
path, _ = synthetic.cached_model(n, 300, 'trained', bits)
configs = [tuple(map(int, c.split('x'))) for c in os.environ.get('TUNE', '1x1,2x4,4x2,4x4,4x8,5x4,8x2,8x4,8x8,10x4,16x4,16x8').split(',')]
rows = torch.arange(n, dtype=torch.int32, device='cuda')
perm = torch.randperm(n, device='cuda').to(torch.int32)
out = torch.empty((n, 300), dtype=torch.float32, device='cuda')
for lanes, waves in configs:
    os.environ['MEMB_HIP_LANES'] = str(lanes); os.environ['MEMB_HIP_WAVES'] = str(waves)
    t0 = time.time()
    reader = memb_amd.Reader(path, device=0)
    info = reader.info()
    topen = time.time() - t0
    line = 'lanes %2d (G=%2d S=%3d) waves %d lds %6d open %.2fs' % (lanes, info['lanes_per_word'], info['segment_symbols'], waves, info['lds_bytes_per_block'], topen)
    for name, r in (('sorted', rows), ('random', perm)):
        for _ in range(3): reader.rows_embedding_device(r, out=out)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a, b in ev:
            a.record(); reader.rows_embedding_device(r, out=out); b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev)
        line += ' | %s min %.3f med %.3f ms %.2f TB/s' % (name, ms[0], ms[len(ms)//2], 2.941*n/2196017/ms[len(ms)//2])
    print(line, flush=True)
    del reader
