"""Per-launch kernel time of the first 400 launches after the GPU has sat idle (model open + a 3 s sleep)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import memb_amd
from memb_amd import synthetic
n = 2196017
path, _ = synthetic.cached_model(n, 300, 'trained', 4)
reader = memb_amd.Reader(path, device=0); reader.info()
rows = torch.arange(n, dtype=torch.int32, device='cuda')
out = torch.empty((n, 300), dtype=torch.float32, device='cuda')
torch.cuda.synchronize()
for idle in (3.0, 0.0, 10.0):
    time.sleep(idle)
    count = 400
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(count)]
    start = time.perf_counter()
    for a, b in ev:
        a.record(); reader.rows_embedding_device(rows, out=out); b.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - start
    ms = [a.elapsed_time(b) for a, b in ev]
    print('after %.0f s idle: launches 0-4 %s | 5-24 avg %.4f | 25-49 avg %.4f | 50-99 avg %.4f | 100-199 avg %.4f | 200-399 avg %.4f | wall/launch %.4f' % (
        idle, ' '.join('%.3f' % x for x in ms[:5]), sum(ms[5:25]) / 20, sum(ms[25:50]) / 25, sum(ms[50:100]) / 50, sum(ms[100:200]) / 100, sum(ms[200:]) / 200, wall / count * 1e3), flush=True)
