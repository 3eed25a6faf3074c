"""Uniform storage, one build (MEMB_PACKAGE_ROOT) per process: the 500 000-word 8-bit dump, the same rows shuffled and
100 000 random rows, each run in for 20 ms and timed as the median of 20 event pairs (or bursts for short kernels).
tools/perf/r4/batch8.sh alternates two builds on one box."""
import os
import sys

ROOT = os.environ.get('MEMB_PACKAGE_ROOT') or os.getcwd()
sys.path.insert(0, os.path.abspath(ROOT))
sys.path.insert(1, os.getcwd())
import numpy as np
import torch

import memb_amd
from memb_amd import synthetic
import bench

count = 500000
path, _ = synthetic.cached_model(count, 300, 'uniform', 8)
reader = memb_amd.Reader(path, device=0)
timer = bench.Timer(torch)
generator = torch.Generator(device='cuda')
generator.manual_seed(3)
rows = torch.arange(count, dtype=torch.int32, device='cuda')
perm = torch.randperm(count, device='cuda', generator=generator).to(torch.int32)
out = torch.empty((count, 300), dtype=torch.float32, device='cuda')
line = os.path.dirname(memb_amd.__file__)
for name, ids in (('dump', rows), ('shuffled', perm), ('100k', perm[:100000].contiguous())):
    target = out[:len(ids)]
    call = lambda: reader.rows_embedding_device(ids, out=target)
    times = timer.launches(call, 20)
    median = times[len(times) // 2]
    if median < 0.2:
        median = timer.bursts(call, 50)[2]
    line += ' | %s %.4f ms' % (name, median)
print(line, reader.info(count)['kernel'], flush=True)
