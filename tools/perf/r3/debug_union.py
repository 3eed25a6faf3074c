import os, sys, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import memb_amd
from memb_amd import synthetic, _memb
lib = ctypes.CDLL(memb_amd.HIP_LIBRARY_PATH); lib.memb_hip_last_error.restype = ctypes.c_char_p
os.makedirs('/tmp/dbg', exist_ok=True)
pa, pb = '/tmp/dbg/a.bin', '/tmp/dbg/b.bin'
synthetic.build_file(pa, 20000, 300, 'trained', 4, seed=1234)
synthetic.build_file(pb, 15000, 300, 'trained', 4, seed=99)
readers = [memb_amd.Reader(pa, device=0), memb_amd.Reader(pb, device=0)]
batch = 90001
rng = np.random.default_rng(1)
ids = [torch.from_numpy(rng.integers(0, c, size=batch).astype(np.uint32).view(np.int32)).cuda() for c in (20000, 15000)]
merged = torch.zeros((batch, 600), dtype=torch.float32, device='cuda')
for persistent, pipeline in ((1, 0), (1, 2), (0, 0)):
    readers[0].set_option('persistent', persistent); readers[0].set_option('pipeline', pipeline)
    ok = _memb.union_rows_to_device([r._impl for r in readers], [t.data_ptr() for t in ids], [0, 300], batch, merged.data_ptr(), 600, 0, False)
    torch.cuda.synchronize()
    print(persistent, pipeline, ok, lib.memb_hip_last_error(), [r.info()['kernel_registers'] for r in readers], flush=True)
