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
