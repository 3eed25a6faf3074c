"""What makes the linear fill fast? (tools/perf/ceilings.hip: store_experiment) -- 2.635 GB written by wavefronts that store
once / S times, after / between sleeps, in blocks of 64 .. 1024 threads; 20 ms run-in, median of 20 launches each."""
import ctypes
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

import bench
import build_native

build_native.build_ceilings()
library = ctypes.CDLL(build_native.CEILINGS_LIBRARY)
library.memb_ceiling_store_experiment.restype = ctypes.c_int
library.memb_ceiling_store_experiment.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
words = 2196017
out = torch.empty((words, 300), dtype=torch.float32, device='cuda')
timer = bench.Timer(torch)
stream = torch.cuda.current_stream().cuda_stream
gigabytes = words * 1200 / 1e9


def run(label, experiment, stores, delay, threads):
    def call():
        status = library.memb_ceiling_store_experiment(out.data_ptr(), words, experiment, stores, delay, threads, stream)
        assert status == 0, status
    times = timer.launches(call, 20)
    median = times[len(times) // 2]
    print('%-78s %.4f ms  %.2f TB/s' % (label, median, gigabytes / median), flush=True)


for threads in (64, 128, 256, 512, 1024):
    run('one store per wavefront, no delay, blocks of %d threads' % threads, 0, 1, 0, threads)
for delay in (1, 2, 5, 10, 20):
    run('one store per wavefront after %4.1f us of sleep, blocks of 256' % (delay * 0.43), 0, 1, delay, 256)
for stores in (2, 3, 5, 10):
    for threads in (64, 256):
        run('%2d consecutive stores per wavefront, no delay, blocks of %d' % (stores, threads), 1, stores, 0, threads)
for stores in (3, 10):
    for delay in (5, 10):
        run('%2d stores per wavefront after %4.1f us of sleep, blocks of 256' % (stores, delay * 0.43), 1, stores, delay, 256)
for stores in (3, 10):
    for delay in (1, 2, 5):
        run('%2d stores per wavefront, %4.1f us of sleep BETWEEN stores, blocks of 256' % (stores, delay * 0.43), 2, stores, delay, 256)


# uniform-storage shapes: one row per wavefront (two stores) against eight rows per wavefront (ten stores), 320-byte records
library.memb_ceiling_launch.restype = ctypes.c_int
library.memb_ceiling_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulonglong,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
generator = torch.Generator(device='cuda')
generator.manual_seed(31)
for rows in (500000, 2196017):
    records = torch.randint(0, 2 ** 31 - 1, (rows, 80), dtype=torch.int32, device='cuda', generator=generator)   # 320 B per row
    ids = torch.randperm(rows, device='cuda', generator=generator).to(torch.int32)
    target = out[:rows]
    for pattern, name in ((8, 'one row per wavefront (2 stores)'), (9, 'eight rows per wavefront (10 stores)')):
        for order, id_pointer in (('consecutive rows', None), ('random rows', ids.data_ptr())):
            def call():
                status = library.memb_ceiling_launch(pattern, target.data_ptr(), rows, records.data_ptr(), None, rows, id_pointer, None, stream, 256)
                assert status == 0, status
            times = timer.launches(call, 20)
            median = times[len(times) // 2]
            if median < 0.2:
                median = timer.bursts(call, 50)[2]
            moved = rows * (1200 + 320) / 1e9
            print('uniform shape, %7d rows, %-38s %-16s %.4f ms  %.2f TB/s of bytes moved' % (rows, name, order, median, moved / median), flush=True)
    del records, ids
