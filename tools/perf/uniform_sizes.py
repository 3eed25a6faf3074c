"""Uniform storage: block kernel (persistent=0) against dequant_uniform_persistent over batch sizes, burst timing, alternating."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import memb_amd
from memb_amd import synthetic
n = 500000
path, _ = synthetic.cached_model(n, 300, 'uniform', 8)
reader = memb_amd.Reader(path, device=0)
generator = torch.Generator(device='cuda'); generator.manual_seed(5)
perm = torch.randperm(n, device='cuda', generator=generator).to(torch.int32)
out = torch.empty((n, 300), dtype=torch.float32, device='cuda')
def burst(rows, view, reps=40):
    for _ in range(5): reader.rows_embedding_device(rows, out=view)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): reader.rows_embedding_device(rows, out=view)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
sizes = [int(x) for x in os.environ.get('US_SIZES', '10000,30000,60000,100000,200000,500000').split(',')]
for count in sizes:
    dump = count == n and os.environ.get('US_DUMP', '1') == '1'
    rows = (torch.arange(n, dtype=torch.int32, device='cuda') if dump else perm[:count].contiguous())
    view = out[:count]
    results = {0: [], 2: []}
    for rnd in range(4):
        for mode in ((0, 2) if rnd % 2 == 0 else (2, 0)):
            reader.set_option('persistent', mode)
            results[mode].append(burst(rows, view))
    reader.set_option('persistent', 1)
    default = burst(rows, view)
    name = reader.info(count)['kernel']
    block, pers = sorted(results[0])[1], sorted(results[2])[1]
    nbytes = count * (4 + 12 + 300 + 1200)
    print('%7d rows%s: block kernel %.4f ms, persistent %.4f ms (%+.1f %%), default %.4f ms (%s) = %.3f of peak' % (
        count, ' (dump)' if dump else '', block, pers, 100 * (pers / block - 1), default, name.split('<')[0], nbytes / default / 1e-3 / 8e12), flush=True)
