"""The memory patterns of tools/perf/ceilings.hip at the batch sizes of the small / medium classes (1 k .. 500 k random rows of
a 2.2 M-row record array, one output buffer reused): what the box does with the bytes of such a batch and no decoder."""
import ctypes
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

import bench
import build_native

build_native.build_ceilings()
library = bench.ceilings_library()
timer = bench.Timer(torch)
stream = torch.cuda.current_stream().cuda_stream
units = torch.cuda.get_device_properties(0).multi_processor_count
rows = 2196017
generator = torch.Generator(device='cuda')
generator.manual_seed(29)
records = torch.randint(0, 2 ** 31 - 1, (rows, 40), dtype=torch.int32, device='cuda', generator=generator)
for words in (1000, 10000, 50000, 100000, 200000, 500000):
    ids = torch.randint(0, rows, (words,), dtype=torch.int32, device='cuda', generator=generator)
    out = torch.empty((words, 300), dtype=torch.float32, device='cuda')
    line = '%7d rows:' % words
    for pattern, name in ((1, 'tiles'), (3, 'tiles + random records'), (6, 'persistent tiles + random records')):
        def call():
            status = library.memb_ceiling_launch(pattern, out.data_ptr(), words, records.data_ptr(), None, rows, ids.data_ptr(), None, stream, units)
            assert status == 0, status
        averages = timer.bursts(call, 100 if words <= 100000 else 30)
        line += '  %s %.2f us' % (name, averages[2] * 1e3)
    print(line, flush=True)
