"""Uniform storage: the kernels against each other on ONE Reader, interleaved, rounds in alternating order
(tile = dequant_uniform_tile, the default; old = the rule before it: LDS-DMA pipeline for large batches, block kernel for
small ones; block = the block kernel always). 20 ms run-in per timing, median of 20 event pairs or bursts."""
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
big = torch.cat([perm, rows, perm.flip(0), rows])[:2000000].contiguous()
out = torch.empty((2000000, 300), dtype=torch.float32, device='cuda')
cases = (('dump 500k', rows), ('shuffled 500k', perm), ('2M rows', big), ('100k', perm[:100000].contiguous()),
         ('10k', perm[:10000].contiguous()), ('1k', perm[:1000].contiguous()))
variants = (('tile', {'persistent': 1, 'tiles_per_wave': 0}), ('old', {'persistent': 1, 'tiles_per_wave': 63}),
            ('block', {'persistent': 0, 'tiles_per_wave': 0}), ('tile2', {'persistent': 1, 'tiles_per_wave': 0}))
results = {}
for rnd in range(4):
    for name, options in (variants if rnd % 2 == 0 else variants[::-1]):
        for key, value in options.items():
            reader.set_option(key, value)
        for case, ids in cases:
            target = out[:len(ids)]
            call = lambda: reader.rows_embedding_device(ids, out=target)
            times = timer.launches(call, 20)
            median = times[len(times) // 2]
            if median < 0.2:
                median = timer.bursts(call, 50)[2]
            results.setdefault((case, name), []).append(median)
for case, _ in cases:
    base = sorted(results[(case, 'tile')])[2]
    print(case)
    for name, _ in variants:
        values = sorted(results[(case, name)])
        print('  %-6s %.4f ms  %+6.2f %%   [%.4f .. %.4f]' % (name, values[2], 100 * (values[2] / base - 1), values[0], values[-1]))
