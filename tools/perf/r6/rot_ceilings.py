"""The memory patterns of tools/perf/ceilings.hip at 100 000 random rows in the HBM regime: K different id sets into K
different output buffers round-robin (K x 120 MB > the 256 MB Infinity Cache), as bench.py takes configs[1]'s figure;
and the patterns of a full dump on the same box (the term table's first line)."""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tools', 'perf'))
import torch

import bench_extras
from bench_support import Timer

library = bench_extras.ceilings_library()
timer = Timer(torch)
stream = torch.cuda.current_stream().cuda_stream
units = torch.cuda.get_device_properties(0).multi_processor_count
rows = 2196017
generator = torch.Generator(device='cuda')
generator.manual_seed(29)
records = torch.randint(0, 2 ** 31 - 1, (rows, 40), dtype=torch.int32, device='cuda', generator=generator)
for words in (60000, 100000, 130000):
    sets = max(4, -(-640000000 // (words * 1200)))
    ids = [torch.randint(0, rows, (words,), dtype=torch.int32, device='cuda', generator=generator) for _ in range(sets)]
    outs = [torch.empty((words, 300), dtype=torch.float32, device='cuda') for _ in range(sets)]
    line = '%7d rows, %d buffers round-robin:' % (words, sets)
    for pattern, name in ((1, 'tiles'), (3, 'tiles + random records'), (6, 'resident tiles + random records'), (11, 'two tiles + random records')):
        turn = [0]

        def call():
            k = turn[0] % sets
            turn[0] += 1
            status = library.memb_ceiling_launch(pattern, outs[k].data_ptr(), words, records.data_ptr(), None, rows, ids[k].data_ptr(), None, stream, units)
            assert status == 0, status
        averages = timer.bursts(call, 15 * sets)
        line += '  %s %.2f us' % (name, averages[2] * 1e3)
    print(line, flush=True)
    del ids, outs
out = torch.empty((rows, 300), dtype=torch.float32, device='cuda')
ids = torch.randperm(rows, device='cuda', generator=generator).to(torch.int32)
line = 'full dump, %d rows:' % rows
for pattern, name in ((0, 'linear fill'), (1, 'tiles'), (2, 'tiles + sequential records'), (3, 'tiles + random records'), (10, 'two tiles + sequential'), (11, 'two tiles + random')):
    def call():
        status = library.memb_ceiling_launch(pattern, out.data_ptr(), rows, records.data_ptr(), None, rows, ids.data_ptr(), None, stream, units)
        assert status == 0, status
    times = timer.launches(call, 20)
    line += '  %s %.4f ms' % (name, times[len(times) // 2])
print(line, flush=True)
