"""Key-order dumps and shuffled full-size batches of the 4-, 6- and 2-bit models with the package under MEMB_PACKAGE_ROOT (default: the
tree): median of 20 launches after a 20 ms run-in, each model's order memory as the batches leave it. One line per model."""
import os
import sys

ROOT = os.environ.get('MEMB_PACKAGE_ROOT') or os.getcwd()
sys.path.insert(0, os.path.abspath(ROOT))
sys.path.insert(0, os.path.join(os.getcwd(), 'tools', 'perf'))
import torch

import memb_amd
from bench_support import Timer
from memb_amd import synthetic

timer = Timer(torch)
line = os.path.dirname(memb_amd.__file__)[-28:]
for words, bits in ((2196017, 4), (1999995, 6), (2196017, 2)):
    path, _ = synthetic.cached_model(words, 300, 'trained', bits)
    reader = memb_amd.Reader(path, device=0)
    rows = torch.arange(words, dtype=torch.int32, device='cuda')
    out = torch.empty((words, 300), dtype=torch.float32, device='cuda')
    generator = torch.Generator(device='cuda')
    generator.manual_seed(5)
    perm = torch.randperm(words, device='cuda', generator=generator).to(torch.int32)
    ms = timer.launches(lambda: reader.rows_embedding_device(rows, out=out), 20)
    w_sorted = reader.info(words)['waves_per_block']
    shuffled = timer.launches(lambda: reader.rows_embedding_device(perm, out=out), 20)
    w_shuffled = reader.info(words)['waves_per_block']
    again = timer.launches(lambda: reader.rows_embedding_device(rows, out=out), 20)
    line += ' | %d-bit sorted %.4f (w%d) shuffled %.4f (w%d) sorted again %.4f' % (bits, ms[10], w_sorted, shuffled[10], w_shuffled, again[10])
    del reader, rows, out, perm
print(line, flush=True)
