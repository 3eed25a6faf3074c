"""Round 5, batch 30: what does the row-number hop (row ids from an array in front of each tile's records) cost a key-order dump?
tools/perf/ceilings.hip: tiles_skeleton with one tile per wavefront (and two), rows computed against rows loaded."""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import torch
import bench, build_native
build_native.build_ceilings()
library = ctypes.CDLL(build_native.CEILINGS_LIBRARY)
library.memb_ceiling_skeleton.restype = ctypes.c_int
library.memb_ceiling_skeleton.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint, ctypes.c_void_p]
words = 2196017
generator = torch.Generator(device='cuda'); generator.manual_seed(29)
out = torch.empty((words, 300), dtype=torch.float32, device='cuda')
records = torch.randint(0, 2 ** 31 - 1, (words, 40), dtype=torch.int32, device='cuda', generator=generator)
in_order = torch.arange(words, device='cuda', dtype=torch.int32)
tables = torch.randint(0, 2 ** 31 - 1, (16384,), dtype=torch.int32, device='cuda', generator=generator)
timer = bench.Timer(torch)
stream = torch.cuda.current_stream().cuda_stream
def run(via_ids, steps, copy_bytes, pad_bytes):
    def call():
        assert library.memb_ceiling_skeleton(out.data_ptr(), words, records.data_ptr(), words, in_order.data_ptr(), 0, via_ids, steps,
                                             tables.data_ptr(), copy_bytes, pad_bytes, stream) == 0
    times = timer.launches(call, 20)
    return times[len(times) // 2]
for repeat in range(2):
    for steps in (1, 2):
        for copy_bytes, pad_bytes, label in ((0, 0, 'bare'), (6144, 14336, 'table copy, 24 wavefronts per CU')):
            a, b = run(0, steps, copy_bytes, pad_bytes), run(1, steps, copy_bytes, pad_bytes)
            print('pass %d  T = %d  %-34s rows computed %.4f ms | rows from an array %.4f ms (%+.1f %%)' % (repeat, steps, label, a, b, 100 * (b / a - 1)), flush=True)
