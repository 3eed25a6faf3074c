"""Where do the 6.5 us of a 1 000-row call go? Bursts of back-to-back calls timed on the host clock (enqueue rate) and with
device events (what the stream sees), for the checked call, Reader.prepared_lookup and an empty-ish kernel through the
same stream; run under `rocprofv3 --kernel-trace --stats` the kernel's own duration is in the trace."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

import memb_amd
from memb_amd import synthetic
import bench

path, _ = synthetic.cached_model(2196017, 300, 'trained', 4)
reader = memb_amd.Reader(path, device=0)
timer = bench.Timer(torch)
for count in (1000, 10000):
    ids = torch.randint(0, 2196017, (count,), dtype=torch.int32, device='cuda')
    out = torch.empty((count, 300), dtype=torch.float32, device='cuda')
    prepared = reader.prepared_lookup(ids, out)
    tiny = torch.empty(64, device='cuda')
    for name, call in (('checked call', lambda: reader.rows_embedding_device(ids, out=out)), ('prepared call', prepared),
                       ('torch fill_ of 64 floats', lambda: tiny.fill_(1.0))):
        call()
        torch.cuda.synchronize()
        launches = 2000
        start = time.perf_counter()
        for _ in range(launches):
            call()
        enqueued = time.perf_counter() - start
        torch.cuda.synchronize()
        finished = time.perf_counter() - start
        device = timer.burst(call, 500) * 1e3
        print('%6d rows  %-26s host enqueue %.2f us per call, host until done %.2f us, device events %.2f us' % (
            count, name, enqueued / launches * 1e6, finished / launches * 1e6, device), flush=True)
