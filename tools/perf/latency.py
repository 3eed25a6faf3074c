"""Latency of small lookups through the Python API."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import memb_amd
from memb_amd import synthetic
path,_=synthetic.cached_model(2196017,300,'trained',4)
r=memb_amd.Reader(path); keys=r.keys(); r['x']
for m in (1, 64, 512, 513, 1024, 2048, 4096, 8192, 20000):
    words=keys[1000:1000+m]
    for _ in range(200): r[words] if m>1 else r[words[0]]
    t=time.perf_counter()
    reps=2000
    for _ in range(reps): (r[words] if m>1 else r[words[0]])
    dt=(time.perf_counter()-t)/reps
    print('n=%4d  %.1f us per call  (%.2f M emb/s)'%(m, dt*1e6, m/dt/1e6), flush=True)
