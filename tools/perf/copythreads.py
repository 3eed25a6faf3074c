"""reader[keys()] and its stages against the number of host threads that empty the pinned ring (MEMB_HIP_COPY_THREADS)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import memb_amd
from memb_amd import synthetic
n = 2196017
path, _ = synthetic.cached_model(n, 300, 'trained', 4)
rng = np.random.default_rng(3)
for threads in (8, 16, 24, 32, 48, 64):
    os.environ['MEMB_HIP_COPY_THREADS'] = str(threads)
    r = memb_amd.Reader(path); keys = r.keys(); r.info()
    rows = r.resolve_rows(keys)
    reused = np.zeros((n, 300), dtype=np.float32)
    sample = [keys[i] for i in rng.integers(0, n, size=100000)]
    best = [1e9] * 4
    for rep in range(4):
        t0 = time.perf_counter(); out = r.rows_embedding(rows); t1 = time.perf_counter(); del out
        t2 = time.perf_counter(); r.rows_embedding_into(rows, reused); t3 = time.perf_counter()
        t4 = time.perf_counter(); full = r.batch_embedding(keys); t5 = time.perf_counter(); del full
        t6 = time.perf_counter(); part = r.batch_embedding(sample); t7 = time.perf_counter(); del part
        best = [min(a, b) for a, b in zip(best, (t1 - t0, t3 - t2, t5 - t4, t7 - t6))]
    print('copy threads %2d: rows->fresh numpy %.1f ms | rows->reused %.1f ms | reader[keys()] %.1f ms | reader[100k words] %.2f ms' % (
        threads, best[0] * 1e3, best[1] * 1e3, best[2] * 1e3, best[3] * 1e3), flush=True)
    del r, reused
