"""Which TILE pattern does this part like? (tools/perf/ceilings.hip: tile_experiment) -- the 2.2 M-row dump's bytes (2.635 GB
written behind 351 MB of 160-byte records) by rows per wavefront, wavefronts per block, a sleep between loads and stores,
and the order of the stores; consecutive and random rows; 20 ms run-in, median of 20 launches each."""
import ctypes
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

import bench
import build_native

build_native.build_ceilings()
library = ctypes.CDLL(build_native.CEILINGS_LIBRARY)
library.memb_ceiling_tile_experiment.restype = ctypes.c_int
library.memb_ceiling_tile_experiment.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_void_p,
                                                 ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
words = 2196017
generator = torch.Generator(device='cuda')
generator.manual_seed(29)
out = torch.empty((words, 300), dtype=torch.float32, device='cuda')
records = torch.randint(0, 2 ** 31 - 1, (words, 40), dtype=torch.int32, device='cuda', generator=generator)
ids = torch.randperm(words, device='cuda', generator=generator).to(torch.int32)
timer = bench.Timer(torch)
stream = torch.cuda.current_stream().cuda_stream
moved = words * 1360 / 1e9


def run(rows_per_tile, waves, delay, order, random):
    def call():
        status = library.memb_ceiling_tile_experiment(out.data_ptr(), words, records.data_ptr(), words, ids.data_ptr() if random else None,
                                                      rows_per_tile, waves, delay, order, stream)
        assert status == 0, status
    times = timer.launches(call, 20)
    return times[len(times) // 2]


print('rows/tile waves/block sleep(us) order        consecutive rows          random rows')
for rows_per_tile in (4, 8, 16):
    for waves in (4, 8, 16):
        for delay in (0, 5):
            for order in (0, 1, 2):
                if rows_per_tile * waves * 160 > 150000:
                    continue
                a = run(rows_per_tile, waves, delay, order, False)
                b = run(rows_per_tile, waves, delay, order, True)
                print('%9d %11d %9.1f %5s   %.4f ms %.2f TB/s    %.4f ms %.2f TB/s' % (
                    rows_per_tile, waves, delay * 0.43, ('asc', 'desc', 'block')[order], a, moved / a, b, moved / b), flush=True)
