"""Round 5, batch 28: T tiles per wavefront in short-lived blocks, with and without the next tile's records in flight during
this tile's stores (tools/perf/ceilings.hip: tiles_chunked), against one tile per wavefront (tile_fill + records): the
2.2 M-row dump's bytes, consecutive and random rows, 20 ms run-in, median of 20 launches, two passes."""
import ctypes
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

import bench
import build_native

build_native.build_ceilings()
library = ctypes.CDLL(build_native.CEILINGS_LIBRARY)
library.memb_ceiling_chunked.restype = ctypes.c_int
library.memb_ceiling_chunked.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_void_p,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
library.memb_ceiling_launch.restype = ctypes.c_int
library.memb_ceiling_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulonglong,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
words = int(os.environ.get('CHUNKED_WORDS', '2196017'))
units = torch.cuda.get_device_properties(0).multi_processor_count
generator = torch.Generator(device='cuda')
generator.manual_seed(29)
out = torch.empty((words, 300), dtype=torch.float32, device='cuda')
records = torch.randint(0, 2 ** 31 - 1, (2196017, 40), dtype=torch.int32, device='cuda', generator=generator)
ids = torch.randperm(2196017, device='cuda', generator=generator)[:words].to(torch.int32).contiguous()
timer = bench.Timer(torch)
stream = torch.cuda.current_stream().cuda_stream


def median(call):
    times = timer.launches(call, 20)
    return times[len(times) // 2]


def reference(pattern):
    def call():
        assert library.memb_ceiling_launch(pattern, out.data_ptr(), words, records.data_ptr(), None, 2196017, ids.data_ptr(), None, stream, units) == 0
    return median(call)


def chunked(waves, steps, prefetch, random, fronts=1):
    def call():
        status = library.memb_ceiling_chunked(out.data_ptr(), words, records.data_ptr(), 2196017, ids.data_ptr() if random else None,
                                              waves, steps, prefetch, fronts, stream)
        assert status == 0, status
    return median(call)


for repeat in range(2):
    print('--- pass %d (%d words)' % (repeat, words))
    sequential, scattered = reference(2), reference(3)
    print('tile_fill + sequential records %.4f ms | + random records %.4f ms' % (sequential, scattered), flush=True)
    print('waves/block  T  prefetch   consecutive rows            random rows')
    for waves in (4, 8):
        for steps, prefetch in ((1, 0), (2, 0), (2, 1), (3, 1), (4, 1)):
            a = chunked(waves, steps, prefetch, False)
            b = chunked(waves, steps, prefetch, True)
            print('%11d %2d %9d   %.4f ms %+6.1f %%        %.4f ms %+6.1f %%' % (
                waves, steps, prefetch, a, 100 * (a / sequential - 1), b, 100 * (b / scattered - 1)), flush=True)

# the same two tiles per wavefront with the PERSISTENT pattern's stride (tile, tile + grid x 4: half a batch apart) instead of a
# block's own run of tiles: tiles_persistent (patterns 5 / 6) launched with as many blocks as give every wavefront T tiles
tiles = (words + 7) // 8
tile_blocks = (tiles + 3) // 4
for steps in (2, 3):
    fake_units = (tile_blocks + steps - 1) // steps // 4 + 1
    def strided(pattern):
        def call():
            assert library.memb_ceiling_launch(pattern, out.data_ptr(), words, records.data_ptr(), None, 2196017, ids.data_ptr(), None, stream, fake_units) == 0
        return median(call)
    a, b = strided(5), strided(6)
    print('T = %d, tiles a grid apart (prefetch): consecutive %.4f ms %+.1f %%   random %.4f ms %+.1f %%' % (
        steps, a, 100 * (a / sequential - 1), b, 100 * (b / scattered - 1)), flush=True)

# one tile per wavefront, but the grid writes F regions of the batch at once (groups of eight blocks take turns between F parts)
for waves in (4, 8):
    for fronts in (2, 4, 8):
        a = chunked(waves, 1, 0, False, fronts)
        b = chunked(waves, 1, 0, True, fronts)
        print('one tile per wavefront, blocks of %d, %d write fronts: consecutive %.4f ms %+.1f %%   random %.4f ms %+.1f %%' % (
            waves, fronts, a, 100 * (a / sequential - 1), b, 100 * (b / scattered - 1)), flush=True)
