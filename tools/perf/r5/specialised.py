"""a tile pattern in which the wavefronts that STORE are not the ones that LOAD
(tools/perf/ceilings.hip: tiles_specialised) against the patterns bench.py reports (tile_fill + records, persistent
tiles + records): the 2.2 M-row dump's bytes, consecutive and random rows, 20 ms run-in, median of 20 launches."""
import ctypes
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

import bench
import build_native

build_native.build_ceilings()
library = ctypes.CDLL(build_native.CEILINGS_LIBRARY)
library.memb_ceiling_specialised.restype = ctypes.c_int
library.memb_ceiling_specialised.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_void_p,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
library.memb_ceiling_launch.restype = ctypes.c_int
library.memb_ceiling_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulonglong,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
words = 2196017
units = torch.cuda.get_device_properties(0).multi_processor_count
generator = torch.Generator(device='cuda')
generator.manual_seed(29)
out = torch.empty((words, 300), dtype=torch.float32, device='cuda')
records = torch.randint(0, 2 ** 31 - 1, (words, 40), dtype=torch.int32, device='cuda', generator=generator)
ids = torch.randperm(words, device='cuda', generator=generator).to(torch.int32)
timer = bench.Timer(torch)
stream = torch.cuda.current_stream().cuda_stream
moved = words * 1360 / 1e9


def median(call):
    times = timer.launches(call, 20)
    return times[len(times) // 2]


def reference(pattern):
    def call():
        assert library.memb_ceiling_launch(pattern, out.data_ptr(), words, records.data_ptr(), None, words, ids.data_ptr(), None, stream, units) == 0
    return median(call)


def specialised(waves, loaders, waves_per_cu, random):
    def call():
        status = library.memb_ceiling_specialised(out.data_ptr(), words, records.data_ptr(), words, ids.data_ptr() if random else None,
                                                  waves, loaders, waves_per_cu, stream, units)
        assert status == 0, status
    return median(call)


for repeat in range(2):
    print('--- pass %d' % repeat)
    base = {}
    for pattern, name in ((1, 'tile_fill (stores only)'), (2, 'tile_fill + sequential records'), (3, 'tile_fill + random records'),
                          (4, 'persistent tiles (stores only)'), (5, 'persistent + sequential records'), (6, 'persistent + random records')):
        base[pattern] = reference(pattern)
        print('%-44s %.4f ms  %.2f TB/s' % (name, base[pattern], (moved if pattern not in (1, 4) else words * 1200 / 1e9) / base[pattern]), flush=True)
    print('waves/block loaders waves/CU   consecutive rows (vs tile_fill + sequential)     random rows (vs tile_fill + random)')
    for waves, loaders in ((4, 1), (4, 2), (8, 1), (8, 2), (8, 4), (16, 2), (16, 4)):
        for waves_per_cu in (16, 32):
            if waves_per_cu < waves:
                continue
            a = specialised(waves, loaders, waves_per_cu, False)
            b = specialised(waves, loaders, waves_per_cu, True)
            print('%11d %7d %8d   %.4f ms %.2f TB/s %+6.1f %%                      %.4f ms %.2f TB/s %+6.1f %%' % (
                waves, loaders, waves_per_cu, a, moved / a, (a / base[2] - 1) * 100, b, moved / b, (b / base[3] - 1) * 100), flush=True)
