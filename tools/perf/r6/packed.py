"""Words that are packed already -> row ids in HBM: call to synchronize, best of 9, 2.2 M words in key order and shuffled."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

import memb_amd
from memb_amd import _memb, synthetic

path, _ = synthetic.cached_model(2196017, 300, 'trained', 4)
reader = memb_amd.Reader(path, device=0)
reader.stage_words()
keys = reader.keys()
rng = np.random.default_rng(41)
order = rng.permutation(len(keys))
line = 'MEMB_PACK_CHUNKS=%s MEMB_PACK_THREADS=%s:' % (os.environ.get('MEMB_PACK_CHUNKS', '-'), os.environ.get('MEMB_PACK_THREADS', '-'))
for name, words in (('key order', keys), ('shuffled', [keys[i] for i in order])):
    encoded = [w.encode('utf-8') for w in words]
    blob = b''.join(encoded)
    starts = np.zeros(len(words) + 1, dtype=np.uint32)
    np.cumsum([len(e) for e in encoded], out=starts[1:])
    rows = torch.empty(len(words), dtype=torch.int32, device='cuda')
    scratch = _memb.WordBatch(0)
    best, fill = 1e9, 1e9
    for _ in range(9):
        torch.cuda.synchronize()
        start = time.perf_counter()
        reader.resolve_packed_device(blob, starts, out=rows)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - start)
        fill = min(fill, _memb._packed_fill_seconds(scratch, blob, starts))
    line += '  %s %.3f ms (fill alone %.3f)' % (name, best * 1e3, fill * 1e3)
print(line, flush=True)
