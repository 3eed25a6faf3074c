"""Host-buffer lookups of the 2.2 M-word dump into a reused result: plain against non-temporal stores in the host
threads' expansion (MEMB_HIP_HOST_STREAMING, read when a Reader is created). Alternating, medians."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import memb_amd
from memb_amd import synthetic
n = int(os.environ.get('HS_WORDS', '2196017'))
path, _ = synthetic.cached_model(n, 300, 'trained', int(os.environ.get('HS_BITS', '4')))
readers = {}
for mode in ('0', '1'):
    os.environ['MEMB_HIP_HOST_STREAMING'] = mode
    readers[mode] = memb_amd.Reader(path, device=0)
rows = np.arange(n, dtype=np.uint32)
out = np.zeros((n, 300), dtype=np.float32)
times = {'0': [], '1': []}
for rep in range(int(os.environ.get('HS_REPS', '12'))):
    for mode in (('0', '1') if rep % 2 == 0 else ('1', '0')):
        t = time.perf_counter(); readers[mode].rows_embedding_into(rows, out); times[mode].append(time.perf_counter() - t)
for mode in ('0', '1'):
    series = sorted(times[mode][2:])
    print('host streaming %s: median %.2f ms, min %.2f, max %.2f  (%d calls; %.1f M embeddings/s)' % (
        mode, 1e3 * series[len(series) // 2], 1e3 * series[0], 1e3 * series[-1], len(series), n / series[len(series) // 2] / 1e6), flush=True)
