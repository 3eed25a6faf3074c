"""Round 5, batch 29: what separates the records pipeline's memory skeleton from the two-tile pattern (tools/perf/ceilings.hip:
tiles_skeleton)? The pattern plus, one at a time and together: a 6 KiB table copy + block barrier per block, LDS padding down to
28 / 24 / 20 resident wavefronts per CU, row numbers through an array."""
import ctypes
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

import bench
import build_native

build_native.build_ceilings()
library = ctypes.CDLL(build_native.CEILINGS_LIBRARY)
library.memb_ceiling_skeleton.restype = ctypes.c_int
library.memb_ceiling_skeleton.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint, ctypes.c_void_p]
library.memb_ceiling_launch.restype = ctypes.c_int
library.memb_ceiling_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulonglong,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
words = 2196017
units = torch.cuda.get_device_properties(0).multi_processor_count
generator = torch.Generator(device='cuda')
generator.manual_seed(29)
out = torch.empty((words, 300), dtype=torch.float32, device='cuda')
records = torch.randint(0, 2 ** 31 - 1, (words, 40), dtype=torch.int32, device='cuda', generator=generator)
shuffled = torch.randperm(words, device='cuda', generator=generator).to(torch.int32)
in_order = torch.arange(words, device='cuda', dtype=torch.int32)
tables = torch.randint(0, 2 ** 31 - 1, (16384,), dtype=torch.int32, device='cuda', generator=generator)
timer = bench.Timer(torch)
stream = torch.cuda.current_stream().cuda_stream


def median(call):
    times = timer.launches(call, 20)
    return times[len(times) // 2]


def reference(pattern):
    def call():
        assert library.memb_ceiling_launch(pattern, out.data_ptr(), words, records.data_ptr(), None, words, shuffled.data_ptr(), None, stream, units) == 0
    return median(call)


def skeleton(random, via_ids, steps, copy_bytes, pad_bytes):
    ids = shuffled if random else in_order
    def call():
        status = library.memb_ceiling_skeleton(out.data_ptr(), words, records.data_ptr(), words, ids.data_ptr(), int(random), int(via_ids), steps,
                                               tables.data_ptr(), copy_bytes, pad_bytes, stream)
        assert status == 0, status
    return median(call)


# LDS per block of four: 5 KiB of slots + copy + pad; 160 KiB per CU: 32 wavefronts up to 20 KiB per block, 28 up to 22.8, 24 up to 26.6, 20 up to 32
variants = [('the two-tile pattern', 0, 0, 0), ('+ 6 KiB table copy and barrier', 0, 6144, 0), ('+ row numbers through an array', 1, 0, 0),
            ('28 wavefronts per CU', 0, 0, 16384), ('24 wavefronts per CU', 0, 0, 20480), ('20 wavefronts per CU', 0, 0, 25600),
            ('copy + ids + 24 per CU (the kernel\'s skeleton)', 1, 6144, 14336), ('copy + ids + 28 per CU', 1, 6144, 10240)]
for repeat in range(2):
    print('--- pass %d' % repeat)
    one_tile, one_tile_random = reference(2), reference(3)
    print('one tile per wavefront: consecutive rows %.4f ms, random rows %.4f ms' % (one_tile, one_tile_random), flush=True)
    for steps in (2, 3):
        for name, via_ids, copy_bytes, pad_bytes in variants:
            a = skeleton(False, via_ids, steps, copy_bytes, pad_bytes)
            b = skeleton(True, via_ids, steps, copy_bytes, pad_bytes)
            print('T = %d  %-48s consecutive %.4f ms %+5.1f %%   random %.4f ms %+5.1f %%' % (
                steps, name, a, 100 * (a / one_tile - 1), b, 100 * (b / one_tile_random - 1)), flush=True)
