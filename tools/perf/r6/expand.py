"""Round 6: would a TWO-KERNEL decoder beat the tile plateau? (a) the output half as a pattern (ceilings.hip patterns 12-14: nibble keys in
memory -> one / two / four float4 per thread, wavefront exits) against the linear fill and the tile patterns; (b) that pattern on one stream
WHILE the real decode kernel without its stores (measurement build, debug = 2) runs on another: what the pair costs together."""
import ctypes
import os
import sys

REPO = os.getcwd()
sys.path.insert(0, os.path.join(REPO, 'build', 'measure'))
sys.path.insert(0, os.path.join(REPO, 'tools', 'perf'))
import torch

import bench_extras
import memb_amd
from bench_support import Timer
from memb_amd import synthetic

library = bench_extras.ceilings_library()
timer = Timer(torch)
units = torch.cuda.get_device_properties(0).multi_processor_count
rows = 2196017
generator = torch.Generator(device='cuda')
generator.manual_seed(29)
records = torch.randint(0, 2 ** 31 - 1, (rows, 40), dtype=torch.int32, device='cuda', generator=generator)
out = torch.empty((rows, 300), dtype=torch.float32, device='cuda')


def pattern(number, stream):
    status = library.memb_ceiling_launch(number, out.data_ptr(), rows, records.data_ptr(), None, rows, None, None, stream, units)
    assert status == 0, status


main = torch.cuda.current_stream().cuda_stream
line = 'patterns, 2.2 M rows:'
for number, name in ((0, 'linear fill'), (1, 'tiles'), (2, 'tiles + sequential records'), (10, 'two tiles + sequential'),
                     (12, 'expand 1 piece/thread'), (13, 'expand 2'), (14, 'expand 4')):
    times = timer.launches(lambda: pattern(number, main), 20)
    line += '  %s %.4f' % (name, times[10])
print(line, flush=True)

path, _ = synthetic.cached_model(rows, 300, 'trained', 4)
reader = memb_amd.Reader(path, device=0)
ids = torch.arange(rows, dtype=torch.int32, device='cuda')
decoded = torch.empty((rows, 300), dtype=torch.float32, device='cuda')
full = timer.launches(lambda: reader.rows_embedding_device(ids, out=decoded), 20)[10]
reader.set_option('debug', 2)
decode_only = timer.launches(lambda: reader.rows_embedding_device(ids, out=decoded), 20)[10]
print('decode_trained: everything %.4f ms, without its stores (debug = 2) %.4f ms' % (full, decode_only), flush=True)
side = torch.cuda.Stream()
for number in (12, 13, 14):
    def pair():
        done = torch.cuda.Event()
        with torch.cuda.stream(side):
            side.wait_stream(torch.cuda.current_stream())
            pattern(number, side.cuda_stream)
            done.record()
        reader.rows_embedding_device(ids, out=decoded)
        torch.cuda.current_stream().wait_event(done)
    times = timer.launches(pair, 20)
    print('decode without stores + expand pattern %d on two streams, both done: %.4f ms' % (number, times[10]), flush=True)
