"""One batch size, one value of the measurement switches (PH_DEBUG; needs MEMB_PACKAGE_ROOT=build/measure), 60 launches:
run under `rocprofv3 --kernel-trace --stats` to read the kernel's own duration with phases switched off (launch gaps and
the host's submission rate excluded, which bound the burst timing of ab3.py below ~7 us)."""
import os, sys
ROOT = os.environ.get('MEMB_PACKAGE_ROOT') or os.getcwd()
sys.path.insert(0, os.path.abspath(ROOT))
import numpy as np, torch
import memb_amd
from memb_amd import synthetic
n = 2196017
path, _ = synthetic.cached_model(n, 300, 'trained', int(os.environ.get('PH_BITS', '4')))
reader = memb_amd.Reader(path, device=0)
count = int(os.environ.get('PH_WORDS', '100000'))
generator = torch.Generator(device='cuda'); generator.manual_seed(5)
rows = torch.randperm(n, device='cuda', generator=generator)[:count].to(torch.int32).contiguous()
out = torch.empty((count, 300), dtype=torch.float32, device='cuda')
for key, value in (('persistent', os.environ.get('PH_PERSISTENT')), ('debug', os.environ.get('PH_DEBUG'))):
    if value not in (None, ''):
        reader.set_option(key, int(value))
for _ in range(60):
    reader.rows_embedding_device(rows, out=out)
torch.cuda.synchronize()
print('kernel', reader.info(count)['kernel'])
