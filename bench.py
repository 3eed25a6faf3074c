#!/usr/bin/env python3
"""Benchmark of the memb batch-lookup hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path over one batch: every word of a synthetic
GloVe-840B-shaped model (2,196,017 words x 300, trained 4-bit) is looked up
once, row ids and the fp32 output resident in HBM (BASELINE.json north_star:
">= 50 % of HBM bandwidth on a 2.2M-word x 300-dim 4-bit batch lookup").

N > 1: one process per GPU. Started by torch.distributed.run (RANK / WORLD_SIZE in the
environment) the script is a rank; started plainly as `python bench.py --gpus N` it first
spawns those N ranks as child processes -- before it makes any GPU call itself -- and relays
rank 0's line. Every rank holds a replica of the model; there is no collective on the data
path. Two measurements:
  * the main line (`scaling: weak`): every rank looks up its own full-size batch -- shards of a
    vocabulary N times as large; value = words all ranks decoded / max-over-ranks time;
  * `strong_scaling`: BASELINE.json configs[3], ONE 2,196,017-word 2-bit dump split N ways as
    memb_amd.sharding.shard_range does (the reference's own split, src/reader.cpp:65-79), per-rank
    kernel time, kernel-only and with the D2H copy of each rank's slice into pinned host memory.
    `--scaling strong` makes that split the main line instead.

At N = 1 the line also carries `configs`: every configuration of BASELINE.json measured in this
run (outside the timed region), each with its kernel time, algorithmic bytes, fraction of the HBM
peak and a sampled bit-compare against the CPU checker.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
HBM_COPY_CEILING_GBPS = 6290.0  # measured float4 copy on MI355X (same guide)
GLOVE_WORDS = 2196017
FASTTEXT_WORDS = 1999995
MISSING = 0xFFFFFFFF
FILL_LAUNCHES = 40
OFF_PATH_SEED = 99   # N(0, 0.4^2) vectors whose 4-bit code has a 9-bit word: byte keys instead of the nibble-key fast path

WORKLOADS = {
    # name: (words, bits, batch) ; batch None = every key (full dump)
    'glove840b-300d-4bit-fullvocab': (GLOVE_WORDS, 4, None),     # north-star headline (SURVEY 8d "H")
    'glove840b-300d-4bit-100k': (GLOVE_WORDS, 4, 100000),        # BASELINE.json configs[1]
    'fasttext2m-300d-6bit-fullvocab': (FASTTEXT_WORDS, 6, None),  # BASELINE.json configs[2]
    'glove840b-300d-2bit-fullvocab': (GLOVE_WORDS, 2, None),     # BASELINE.json configs[3]
    'small-4bit': (50000, 4, None),                              # quick functional run
    # the two below are not single trained models: main() takes their step from special_workload()
    'union-concat-500k': (GLOVE_WORDS, 4, 500000),               # BASELINE.json configs[4] as the timed step (profiling runs)
    'uniform-8bit-500k': (500000, 8, None),                      # uniform storage dump as the timed step (profiling runs)
}
SPECIAL_WORKLOADS = ('union-concat-500k', 'uniform-8bit-500k')
STRONG_WORKLOAD = 'glove840b-300d-2bit-fullvocab'


def parse_args():
    parser = argparse.ArgumentParser()
    parser.add_argument('--gpus', type=int, default=1)
    parser.add_argument('--steps', type=int, default=20)
    parser.add_argument('--warmup', type=int, default=5)
    parser.add_argument('--workload', default='glove840b-300d-4bit-fullvocab', choices=sorted(WORKLOADS))
    parser.add_argument('--scaling', default='weak', choices=('weak', 'strong'),
                        help='strong: the main line is ONE batch split over the ranks (default workload then: configs[3])')
    parser.add_argument('--cache-dir', default=os.environ.get('MEMB_BENCH_CACHE', '/tmp/memb_amd_bench'))
    parser.add_argument('--no-cpu-baseline', action='store_true')
    parser.add_argument('--no-configs', action='store_true',
                        help='skip the per-configuration array and the strong-scaling leg (profiling runs: only the timed kernel)')
    parser.add_argument('--no-ceilings', action='store_true', help='skip roofline.box_ceilings (tools/perf/ceilings.hip patterns)')
    parser.add_argument('--no-live-traffic', action='store_true',
                        help='roofline.traffic from profiles/hbm_traffic.json instead of two rocprofv3 --pmc child passes of this run')
    parser.add_argument('--small', action='store_true', help='shrink every model to 50 000 words (plumbing rehearsal)')
    parser.add_argument('--host-writer', action='store_true',
                        help='write the synthetic models with the host writer (default: memb_amd.Builder(device=...), same bytes)')
    parser.add_argument('--dry-launch', action='store_true',
                        help='--gpus N without a launcher: build, check, print the launch command and what the parent saw; start nothing')
    return parser.parse_args()


# --------------------------------------------------------------------------------------------
# launching the ranks
# --------------------------------------------------------------------------------------------

def kfd_gpu_count(topology='/sys/class/kfd/kfd/topology/nodes'):
    """GPUs of this node as the kernel driver lists them: topology nodes with SIMDs (CPU nodes have
    simd_count 0). Reads sysfs only -- no HIP, no torch. None when the driver's tree is not there."""
    try:
        nodes = os.listdir(topology)
    except OSError:
        return None
    count = 0
    for node in nodes:
        try:
            with open(os.path.join(topology, node, 'properties')) as f:
                for line in f:
                    fields = line.split()
                    if len(fields) == 2 and fields[0] == 'simd_count' and int(fields[1]) > 0:
                        count += 1
        except (OSError, ValueError):
            continue   # a node this user may not read is not a GPU this user can run on
    return count


def hip_runtime_mapped():
    """Has this process mapped a HIP / HSA runtime library (the first step of touching the GPU)?"""
    with open('/proc/self/maps') as f:
        return any('libamdhip64' in line or 'libhsa-runtime64' in line for line in f)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children of this process.

    This parent never touches the GPU: it imports neither torch nor memb_amd (both map libamdhip64),
    builds the native code with hipcc (a compiler run, no device) and counts GPUs from the kernel
    driver's sysfs tree. It checks /proc/self/maps for a HIP runtime before it starts the children and
    hands what it saw to rank 0 (`launcher` in the JSON line), so nothing that has initialised the
    GPU is ever the parent of, or replaced by, another GPU program. The ranks are ordinary child
    processes; their output passes through."""
    import socket
    import build_native
    build_native.build_all()
    rehearsal = os.environ.get('MEMB_BENCH_REHEARSAL') in ('1', 'cpu')
    available = kfd_gpu_count()
    # refuse only what is certain: no compute driver at all, or a readable topology with fewer GPUs than asked for
    # (a topology this user cannot read counts nothing: the ranks then find out for themselves)
    if not rehearsal and (available is None or 0 < available < args.gpus):
        raise SystemExit('--gpus {}: this node has {} GPU(s) ({})'.format(
            args.gpus, available or 0, 'kfd topology' if available is not None else 'no /sys/class/kfd: no amdgpu compute driver'))
    with socket.socket() as probe:
        probe.bind(('127.0.0.1', 0))
        port = probe.getsockname()[1]
    command = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + \
        [argument for argument in sys.argv[1:] if argument != '--dry-launch']
    launcher = {'started_by': 'bench.py (child processes)', 'parent_mapped_hip_runtime': hip_runtime_mapped(),
                'gpus_in_kfd_topology': available, 'parent_imported_torch': 'torch' in sys.modules}
    if launcher['parent_mapped_hip_runtime']:
        raise SystemExit('bench.py: the launching process has a HIP runtime mapped; refusing to start GPU ranks from it')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MEMB_BENCH_PREBUILT='1', MEMB_BENCH_LAUNCHER=json.dumps(launcher))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    if args.dry_launch:
        print(json.dumps({'launcher': launcher, 'command': command}))
        return 0
    return subprocess.run(command, env=env).returncode


# --------------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------------

def install_host_stand_ins(torch, memb_amd):
    """MEMB_BENCH_REHEARSAL=cpu: the multi-rank PLUMBING of this script on a machine without a GPU (8 ranks in
    the CPU test suite: launcher, rendezvous, barriers, the max-over-ranks reduce, all_gather_object of the
    per-rank summaries, the strong-scaling split, the JSON line). Everything that would touch the device is
    replaced by a host stand-in -- wall-clock `events`, tensors in host memory, a Reader on the product's host
    path (device='cpu', the reference's own serial / threaded decode restated) -- so the numbers such a run
    prints are NOT measurements of anything; the line says so in `rehearsal`. Never set by the driver."""
    import numpy as np

    class Event:
        def __init__(self, enable_timing=True):
            self.at = 0.0

        def record(self):
            self.at = time.perf_counter()

        def elapsed_time(self, other):
            return max((other.at - self.at) * 1e3, 1e-6)

    class Stream:
        cuda_stream = 0

    torch.cuda.Event = Event
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.current_stream = lambda *a, **k: Stream()
    torch.Tensor.cuda = lambda self, *a, **k: self

    def on_host(function):
        def wrapped(*args, **kwargs):
            if str(kwargs.get('device', '')).startswith('cuda'):
                kwargs['device'] = 'cpu'
            kwargs.pop('pin_memory', None)
            return function(*args, **kwargs)
        return wrapped

    for name in ('empty', 'zeros', 'full', 'tensor', 'arange'):
        setattr(torch, name, on_host(getattr(torch, name)))

    product_reader = memb_amd.Reader

    class HostReader(product_reader):
        def __init__(self, filename, num_threads=0, device=None, **kwargs):
            super().__init__(filename, num_threads, device='cpu', **kwargs)

        def rows_embedding_device(self, rows, out=None, col_off=0, accumulate=False, divisor=0.0, order=None):
            ids = np.ascontiguousarray(rows.numpy()).view(np.uint32)
            if out is None:
                out = torch.empty((len(ids), self.dim), dtype=torch.float32)
            self.rows_embedding_into(ids, out.numpy(), col_off)
            return out

    memb_amd.Reader = HostReader


class Timer:
    """Per-launch device time from HIP events on torch's current stream (the stream the
    kernels are enqueued on: Reader.rows_embedding_device passes it into the C ABI)."""

    def __init__(self, torch):
        self.torch = torch

    def launches(self, call, count, run_in_ms=20.0):
        """Sorted per-launch times of `count` launches that follow ~run_in_ms of the same launches without a
        gap (the part's power state needs that long to settle after an idle gap: tools/perf/ramp.py)."""
        torch = self.torch
        call()
        torch.cuda.synchronize()
        begin, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        begin.record()
        call()
        end.record()
        torch.cuda.synchronize()
        one = max(begin.elapsed_time(end), 1e-3)
        events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(count)]
        for _ in range(max(3, min(2000, int(run_in_ms / one) + 1))):
            call()
        for begin, end in events:
            begin.record()
            call()
            end.record()
        torch.cuda.synchronize()
        return sorted(begin.elapsed_time(end) for begin, end in events)

    def burst(self, call, count, run_in_ms=20.0):
        """Average time per launch of `count` launches enqueued back to back between ONE pair of events (after the
        same run-in). An event pair around every single launch adds 4-5 us of its own -- 15 % of a 100 000-word
        batch (rocprofv3 kernel duration 29.6 us, per-launch events 35.2 us: profiles/r03_100k_*) -- so short
        kernels are quoted this way; the figure includes the ~1.5 us gap between dependent launches."""
        torch = self.torch
        call()
        torch.cuda.synchronize()
        begin, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        begin.record()
        call()
        end.record()
        torch.cuda.synchronize()
        one = max(begin.elapsed_time(end), 1e-3)
        for _ in range(max(3, min(2000, int(run_in_ms / one) + 1))):
            call()
        begin.record()
        for _ in range(count):
            call()
        end.record()
        torch.cuda.synchronize()
        return begin.elapsed_time(end) / count

    def graph_burst(self, call, count, run_in_ms=20.0):
        """As `burst`, with the `count` launches captured into ONE HIP graph and the replay timed: the host's launch rate (5-8 us
        per Python call on some boxes) is then out of a figure that is about a 5 us kernel. Returns (ms per launch, 'graph'),
        or burst's figure and 'eager' where the capture fails."""
        torch = self.torch
        try:
            call()
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                call()                      # (on the capture stream once, outside the capture)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                for _ in range(count):
                    call()
            begin, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(max(2, int(run_in_ms / 0.5 / max(count, 1)) + 1)):
                graph.replay()
            samples = []
            for _ in range(5):
                begin.record()
                graph.replay()
                end.record()
                torch.cuda.synchronize()
                samples.append(begin.elapsed_time(end) / count)
            return sorted(samples)[len(samples) // 2], 'graph'
        except Exception:   # noqa: BLE001 -- a runtime that cannot capture is no reason to lose the line
            torch.cuda.synchronize()
            return self.burst(call, count, run_in_ms), 'eager'

    def bursts(self, call, count, repeats=5, run_in_ms=20.0):
        """`repeats` bursts (each as `burst`, enqueued without a gap after one run-in): sorted averages per launch.
        Minimum, median and average of a short kernel then all come from ONE method."""
        torch = self.torch
        call()
        torch.cuda.synchronize()
        begin, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        begin.record()
        call()
        end.record()
        torch.cuda.synchronize()
        one = max(begin.elapsed_time(end), 1e-3)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(repeats + 1)]
        for _ in range(max(3, min(2000, int(run_in_ms / one) + 1))):
            call()
        marks[0].record()
        for k in range(repeats):
            for _ in range(count):
                call()
            marks[k + 1].record()
        torch.cuda.synchronize()
        return sorted(marks[k].elapsed_time(marks[k + 1]) / count for k in range(repeats))


CEILING_PATTERNS = (
    # (pattern number of tools/perf/ceilings.hip, key, what it is)
    (0, 'linear_fill', 'one 16-byte store per thread, wavefront exits: the best write pattern of this part'),
    (1, 'tile_fill', 'one 9600-byte tile (8 rows) per wavefront, then exit: the stores of decode_trained, nothing else'),
    (2, 'tile_fill_sequential_records', 'tile_fill + the tile\'s eight 160-byte row records read first, consecutive rows (a key-order dump); stored values depend on the loaded bytes'),
    (3, 'tile_fill_random_records', 'tile_fill + eight 160-byte records at random rows (two 128-byte lines each)'),
    (4, 'persistent_tile_fill', '16 resident wavefronts per CU walk the tiles, stores only: the store pattern of a persistent kernel (decode_records_persistent; rounds 1-3: the general pipeline)'),
    (5, 'persistent_tile_fill_sequential_records', 'persistent_tile_fill + sequential records, next tile\'s loads in flight during the stores'),
    (6, 'persistent_tile_fill_random_records', 'persistent_tile_fill + random records, same prefetch'),
    (10, 'two_tiles_sequential_records', 'pattern 5 with a grid of tiles / 2 wavefronts instead of a resident one: two tiles per wavefront half a batch apart, the second '
                                         'tile\'s records in flight during the first tile\'s stores, then exit (round 5, batch 28: the fastest tile pattern found so far)'),
    (11, 'two_tiles_random_records', 'the same behind random records'),
)


def ceilings_library():
    """tools/perf/libmemb_ceilings.so (measurement only; built by build_native.py), or None."""
    path = os.path.join(REPO, 'tools', 'perf', 'libmemb_ceilings.so')
    if not os.path.exists(path):
        return None
    library = ctypes.CDLL(path)
    library.memb_ceiling_launch.restype = ctypes.c_int
    library.memb_ceiling_launch.argtypes = [
        ctypes.c_int, ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulonglong,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    return library


def box_ceilings(torch, timer, out, words, launches=20, union=None, patterns=None):
    """What THIS box does with the decoder's memory pattern and no decoder (tools/perf/ceilings.hip): the
    2.635 GB of a 2.2 M-word dump written as a linear fill, as the decoder's tiles, and as tiles behind the
    reads a decoder of 160-byte row records makes -- same output buffer, same 20 ms run-in, per-launch HIP
    events on the launch stream (median). `union` = (merged output, words): the 500 000 x 600 union shape.
    None where the library is not built."""
    library = ceilings_library()
    if library is None:
        return None
    device = out.device
    units = torch.cuda.get_device_properties(device).multi_processor_count
    rows = int(words)
    generator = torch.Generator(device=device)
    generator.manual_seed(29)
    records = torch.randint(0, 2 ** 31 - 1, (rows, 40), dtype=torch.int32, device=device, generator=generator)   # 160 B per row
    ids = torch.randperm(rows, device=device, generator=generator).to(torch.int32)
    stream = torch.cuda.current_stream().cuda_stream
    result = {'what': 'this box, the decoder\'s memory pattern without a decoder (tools/perf/ceilings.hip): {} rows x 300 floats into the '
                      'bench output buffer, 160-byte row records, run in for 20 ms, median of {} launches (HIP events)'.format(rows, launches)}
    out_bytes = 4.0 * rows * 300

    def run(pattern, target, count, first_ids, records2=None, ids2=None):
        def call():
            status = library.memb_ceiling_launch(
                pattern, target.data_ptr(), count, records.data_ptr(), records2.data_ptr() if records2 is not None else None, rows,
                first_ids.data_ptr(), ids2.data_ptr() if ids2 is not None else None, stream, units)
            if status != 0:
                raise RuntimeError('memb_ceiling_launch({}) failed: hipError {}'.format(pattern, status))
        times = timer.launches(call, launches)
        return times[len(times) // 2]

    for pattern, key, what in CEILING_PATTERNS:
        if patterns is not None and pattern not in patterns:
            continue
        ms = run(pattern, out, rows, ids)
        reads = 0.0 if pattern in (0, 1, 4) else 160.0 * rows
        result[key] = {'what': what, 'ms': ms, 'bytes_moved_GBps': (out_bytes + reads) / (ms * 1e-3) / 1e9}
    if union is not None and (patterns is None or 7 in patterns):
        merged, batch = union
        ids_a = torch.randint(0, rows, (batch,), dtype=torch.int32, device=device, generator=generator)
        ids_b = torch.randint(0, rows, (batch,), dtype=torch.int32, device=device, generator=generator)
        ids_a[torch.rand(batch, device=device, generator=generator) < 0.25] = -1
        ids_b[torch.rand(batch, device=device, generator=generator) < 0.25] = -1
        records2 = torch.randint(0, 2 ** 31 - 1, (rows, 40), dtype=torch.int32, device=device, generator=generator)
        ms = run(7, merged, batch, ids_a, records2, ids_b)
        result['union_tile_fill_random_records'] = {
            'what': 'the union shape: {} merged rows of 600 floats, a tile = 4 rows, eight 160-byte records at random rows of two arrays, 25 % of them absent (not loaded)'.format(batch),
            'ms': ms, 'bytes_moved_GBps': (4.0 * batch * 600 + 160.0 * 2 * 0.75 * batch) / (ms * 1e-3) / 1e9}
    if patterns is None or 9 in patterns:
        uniform_rows = min(500000, rows)
        wide = torch.randint(0, 2 ** 31 - 1, (uniform_rows, 80), dtype=torch.int32, device=device, generator=generator)   # 320 B per row
        target = out[:uniform_rows]

        def call():
            status = library.memb_ceiling_launch(9, target.data_ptr(), uniform_rows, wide.data_ptr(), None, uniform_rows, None, None, stream, units)
            if status != 0:
                raise RuntimeError('memb_ceiling_launch(9) failed: hipError {}'.format(status))
        times = timer.launches(call, launches)
        ms = times[len(times) // 2]
        result['uniform_tile_fill_sequential_records'] = {
            'what': 'the uniform storage\'s shape: {} rows of 300 floats, eight rows per wavefront, 320-byte row records of consecutive rows read first (dequant_uniform_tile\'s loads and stores)'.format(uniform_rows),
            'ms': ms, 'bytes_moved_GBps': uniform_rows * 1520.0 / (ms * 1e-3) / 1e9}
    return result


def algorithmic_bytes(library, reader, rows_host):
    """per word the row id, the index entry, the compressed payload and the fp32 row."""
    import numpy as np
    rows_host = np.ascontiguousarray(rows_host, dtype=np.uint32)
    if os.environ.get('MEMB_BENCH_REHEARSAL') == 'cpu':
        return int(len(rows_host)) * (8 + 4 * reader.dim)   # (no device context to ask; the rehearsal's numbers mean nothing)
    total = ctypes.c_uint64(0)
    status = library.memb_hip_algorithmic_bytes(
        ctypes.c_void_p(reader._impl.context_handle()), rows_host.ctypes.data_as(ctypes.c_void_p),
        ctypes.c_size_t(len(rows_host)), ctypes.byref(total))
    if status != 0:
        raise RuntimeError('memb_hip_algorithmic_bytes failed')
    return total.value


def sampled_parity(path, rows_host, got_rows, sample=20000, seed=5):
    """Bit-compare `sample` rows of a device result (a callable index array -> numpy rows) with the CPU checker."""
    import numpy as np
    import oracle
    rng = np.random.default_rng(seed)
    count = len(rows_host)
    picks = np.sort(rng.choice(count, size=min(sample, count), replace=False))
    expected = oracle.OracleReader(path, os.cpu_count() or 1).rows_embedding(np.ascontiguousarray(rows_host[picks]))
    got = got_rows(picks)
    same = np.array_equal(np.ascontiguousarray(got).view(np.uint32), expected.view(np.uint32))
    return ('bit-exact ({} sampled rows)'.format(len(picks))) if same else 'MISMATCH'


def batch_rows(count, batch, np):
    if batch is None:
        return np.arange(count, dtype=np.uint32)   # batch = keys(): rows in sorted-word order
    rng = np.random.default_rng(11)
    rows = rng.integers(0, count, size=batch).astype(np.uint32)
    rows[rng.integers(0, batch, size=batch // 100)] = MISSING   # 1 % misses
    return rows


def cpu_baseline(path, rows_host, dim):
    """CPU decode of the same batch on this box's host cores, pre-resolved rows, decode only.

    kind "reference": the reference's own HuffmanTableDecoder + centroid gather (oracle/_ref, compiled
    from /root/reference/src in the build container with the reference's -O3; the prebuilt library
    travels with the tree), split over threads as Reader::batchEmbeddingToBuffer splits a batch.
    kind "port": oracle/memb_oracle.c, when oracle/_ref is not there. Either way the output is also
    the parity check of the timed GPU result.
    """
    import numpy as np
    import oracle
    cores = os.cpu_count() or 1
    reader = oracle.OracleReader(path, cores)
    kind = 'port'
    decode = lambda rows, out, threads: reader.rows_embedding(rows, out=out, num_threads=threads)
    if oracle.reference_available() and reader.trained_view() is not None:
        reference = oracle.ReferenceDecoder(reader)
        kind = 'reference'
        decode = lambda rows, out, threads: reference.rows_embedding(rows, out=out, num_threads=threads)
    sample = rows_host
    out = np.empty((len(sample), dim), dtype=np.float32)
    best = float('inf')
    deadline = time.time() + 6.0
    passes = 0
    while passes < 3 or time.time() < deadline:
        start = time.time()
        decode(sample, out, cores)
        best = min(best, time.time() - start)
        passes += 1
    single = sample[:min(len(sample), 100000)]
    start = time.time()
    decode(single, out[:len(single)], 1)
    single_rate = len(single) / (time.time() - start)
    if kind == 'reference':
        # the restatement must agree with the reference on this batch too
        port = reader.rows_embedding(single, num_threads=cores)
        if not np.array_equal(port.view(np.uint32), out[:len(single)].view(np.uint32)):
            raise SystemExit('oracle/memb_oracle.c and oracle/_ref disagree')
    return {
        'value': len(sample) / best,
        'unit': 'embeddings/s',
        'cores': cores,
        'kind': kind,
        'sample': '{} pre-resolved rows of the same batch, decode only ({}), all host threads, best of {} passes (~6 s); single thread: {:.0f} embeddings/s on {} rows'.format(
            len(sample),
            "reference's HuffmanTableDecoder, oracle/_ref" if kind == 'reference' else 'oracle/memb_oracle.c',
            passes, single_rate, len(single)),
    }, out


def host_api_timings(reader, path, rows_host):
    """reader[words] -> numpy for the whole batch and for 100 000 of its words, best of 3, with the
    stages of the whole-batch call timed one by one; and the CPU restatement's Reader.batch_embedding
    (word search + decode, all host threads) on the 100 000."""
    import numpy as np
    import oracle
    keys = reader.keys()
    words = [keys[r] if r < len(keys) else 'not a word' for r in rows_host]
    rng = np.random.default_rng(3)
    sample = [words[i] for i in rng.integers(0, len(words), size=min(100000, len(words)))]

    def best_of(call, repeats=3):
        best = float('inf')
        for _ in range(repeats):
            start = time.perf_counter()
            result = call()
            best = min(best, time.perf_counter() - start)
            del result
        return best

    whole = best_of(lambda: reader.batch_embedding(words))
    part = best_of(lambda: reader.batch_embedding(sample))
    # the stages of the whole-batch call, each on its own
    search = best_of(lambda: reader.resolve_rows(words))
    resolved = reader.resolve_rows(words)
    fresh = best_of(lambda: reader.rows_embedding(resolved))
    reused = np.empty((len(words), reader.dim), dtype=np.float32)
    reused[:] = 0   # pages touched
    into = best_of(lambda: reader.rows_embedding_into(resolved, reused))
    del reused
    checker = oracle.OracleReader(path, os.cpu_count() or 1)
    cpu_part = best_of(lambda: checker.batch_embedding(sample))
    return {
        'note': 'words in, numpy float32 out (word search -- on the device from 4096 words on: memb_hip_decode_words --, PCIe, host memory, '
                'result allocation included); never part of value. The breakdown times the HOST search and the decode of resolved rows on their own',
        'batch_words': len(words),
        'batch_seconds': whole,
        'batch_embeddings_per_s': len(words) / whole,
        'batch_breakdown_seconds': {
            'word_search (resolve_rows, host threads)': search,
            'rows -> fresh numpy result (kernel, PCIe, host expansion, first touch of the result pages)': fresh,
            'rows -> reused, already touched result': into,
            'first touch of the result pages (difference of the two)': fresh - into,
        },
        'sample_words': len(sample),
        'sample_seconds': part,
        'sample_embeddings_per_s': len(sample) / part,
        'cpu_port_sample_seconds': cpu_part,
        'cpu_port_sample_embeddings_per_s': len(sample) / cpu_part,
    }


def word_search_timings(reader, path, torch, np, repeats=5):
    """Word -> row, the step in front of the path (reference src/trained_compression.cpp:115-125,
    python/memb_bindings.cpp:54-63): the host search (hash index on pooled threads, Reader.resolve_rows) against the
    device search (Reader.resolve_rows_device: the words written once into pinned memory by pooled threads, read over
    PCIe and looked up by resolve_words; timed from the call to the row ids being in HBM, synchronize included) on the
    same Python lists, every answer compared with the host search and a sample with the CPU checker's binary search."""
    import oracle
    from memb_amd import _memb
    keys = reader.keys()
    count = len(keys)
    rng = np.random.default_rng(41)
    order = rng.permutation(count)
    hundred = [keys[i] for i in rng.integers(0, count, size=min(100000, count))]
    for i in range(0, len(hundred), 100):
        hundred[i] = hundred[i] + '?'   # 1 % misses
    batches = (('all keys, key order', keys), ('all keys, shuffled', [keys[i] for i in order]), ('100 000 random words, 1 % misses', hundred))
    scratch = _memb.WordBatch(reader.device)
    checker = oracle.OracleReader(path)
    result = {'what': 'word -> row for Python lists of str: host = Reader.resolve_rows (hash index, pooled threads), device = '
                      'Reader.resolve_rows_device (strings -> pinned memory on pooled threads, resolve_words reads them over PCIe; call to '
                      'rows-in-HBM incl. synchronize); best of {} each'.format(repeats),
              'index': {k: reader.info()[k] for k in ('word_index_bytes', 'word_index_slots', 'word_index_keys')}, 'batches': []}

    def best(call):
        times = []
        for _ in range(repeats):
            start = time.perf_counter()
            call()
            times.append(time.perf_counter() - start)
        return min(times)

    for name, words in batches:
        rows = torch.empty(len(words), dtype=torch.int32, device='cuda')

        def device():
            reader.resolve_rows_device(words, out=rows)
            torch.cuda.synchronize()

        device()
        expected = reader.resolve_rows(words)
        agree = bool(np.array_equal(rows.cpu().numpy().view(np.uint32), expected))
        picks = rng.choice(len(words), size=min(20000, len(words)), replace=False)
        sample = [words[i] for i in picks]
        agree_checker = bool(np.array_equal(checker.resolve_rows(sample), expected[picks]))
        host_s = best(lambda: reader.resolve_rows(words))
        device_s = best(device)
        fill_s = best(lambda: _memb._word_fill_seconds(scratch, words))
        begin, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        stream = torch.cuda.current_stream().cuda_stream
        kernel_ms = []
        for _ in range(repeats):
            begin.record()
            reader._impl.resolve_batch_to_device(reader._word_batch, rows.data_ptr(), stream)
            end.record()
            torch.cuda.synchronize()
            kernel_ms.append(begin.elapsed_time(end))
        result['batches'].append({
            'batch': name, 'words': len(words), 'host_ms': host_s * 1e3, 'device_ms': device_s * 1e3, 'speedup': host_s / device_s,
            'device_breakdown_ms': {'strings -> pinned memory alone (no lookup)': fill_s * 1e3,
                                    'resolve_words over the whole batch alone (reads the words over PCIe)': min(kernel_ms)},
            'words_per_s_device': len(words) / device_s,
            'parity': ('device == host search on every word; host == CPU checker (lower_bound + strcmp) on {} sampled words'.format(len(sample))
                       if agree and agree_checker else 'MISMATCH'),
        })
    return result


def live_traffic(workload, kernel_name, cache_dir, timeout=90):
    """HBM bytes per launch of the timed kernel from the PMC counters, collected in THIS run: two child processes,
    `rocprofv3 --pmc FETCH_SIZE` and `rocprofv3 --pmc WRITE_SIZE` (separate passes, no trace domain: MI355X_MICROARCH.md),
    each over `python3 bench.py --workload <this one> --steps 3 --warmup 1` with everything but the timed step switched off;
    mean over the launches of the kernel `roofline.kernel` names. FETCH_SIZE (KB) counts 64 B per 128-byte request of wide
    reads on gfx950 and is doubled, WRITE_SIZE (KB) is exact. Returns (bytes, description) or (None, why not)."""
    import csv
    import glob
    import shutil
    import tempfile
    profiler = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if profiler is None:
        return None, 'rocprofv3 not found'
    # this process is itself being profiled (rocprofv3 ... -- python3 bench.py): a profiler inside a profiler is asking for
    # trouble, and whoever runs that has the counters anyway
    if any(name.startswith(('ROCPROF', 'ROCPROFILER', 'ROCP_')) for name in os.environ) or 'rocprofiler' in os.environ.get('LD_PRELOAD', ''):
        return None, 'this run is itself under a profiler'
    readings = {}
    scratch = tempfile.mkdtemp(prefix='memb_bench_pmc_', dir='/tmp')
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            out = os.path.join(scratch, counter)
            command = [profiler, '--pmc', counter, '--output-format', 'csv', '-d', out, '-o', 'pmc', '--',
                       sys.executable, os.path.abspath(__file__), '--workload', workload, '--steps', '3', '--warmup', '1',
                       '--no-configs', '--no-cpu-baseline', '--no-ceilings', '--no-live-traffic', '--cache-dir', cache_dir]
            env = dict(os.environ, TMPDIR='/tmp', MEMB_BENCH_PREBUILT='1')
            try:
                done = subprocess.run(command, cwd='/tmp', env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
            except subprocess.TimeoutExpired:
                return None, 'rocprofv3 --pmc {} pass timed out after {} s'.format(counter, timeout)
            if done.returncode != 0:
                return None, 'rocprofv3 --pmc {} pass failed (exit code {})'.format(counter, done.returncode)
            values = []
            for path in glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True):
                with open(path) as f:
                    for row in csv.DictReader(f):
                        if row.get('Counter_Name') == counter and kernel_name in row.get('Kernel_Name', ''):
                            values.append(float(row['Counter_Value']))
            if not values:
                return None, 'no {} readings for {}'.format(counter, kernel_name)
            readings[counter] = (sum(values) / len(values), len(values))
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    read_bytes = 2.0 * readings['FETCH_SIZE'][0] * 1024
    write_bytes = readings['WRITE_SIZE'][0] * 1024
    return int(round(read_bytes + write_bytes)), (
        'live: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE child passes of this run (mean of {} / {} launches of the kernel; '
        'reads = 2 x FETCH_SIZE KB = {:.0f} B, writes = WRITE_SIZE KB = {:.0f} B)'.format(
            readings['FETCH_SIZE'][1], readings['WRITE_SIZE'][1], read_bytes, write_bytes))


def prebuild_models(synthetic, models, workers=3):
    """Write the synthetic models this run needs and the box does not have yet, a few at a time (the builder
    releases the GIL while it works; the device writer streams every model's vectors to the GPU as they are drawn).
    Returns the wall time spent."""
    from concurrent.futures import ThreadPoolExecutor
    start = time.time()
    missing = [m for m in models if not os.path.exists(synthetic.cached_model_path(*m))]
    if missing:
        with ThreadPoolExecutor(max_workers=min(workers, len(missing))) as pool:
            list(pool.map(lambda m: synthetic.cached_model(*m), missing))
    return time.time() - start if missing else 0.0


def open_reader(memb_amd, path, device, batch_words=0):
    """Open + stage on this rank's GPU: the model (info() stages it) and the word -> row index (the keys and the hash
    table over them, memb_hip_ctx_stage_words), so that `reader_open_s` and `device_bytes` are what a rank that serves
    words -- not only row ids -- pays."""
    start = time.time()
    reader = memb_amd.Reader(path, device=device)
    reader.info(batch_words)   # stages the model to HBM
    if os.environ.get('MEMB_BENCH_REHEARSAL') != 'cpu':
        reader.stage_words()
    info = reader.info(batch_words)   # the kernel named is the one a batch of that size runs
    return reader, info, time.time() - start


# --------------------------------------------------------------------------------------------
# one configuration of BASELINE.json: kernel time, algorithmic bytes, parity sample
# --------------------------------------------------------------------------------------------

def measure_config(name, what, reader, path, rows_host, timer, library, torch, np, launches=15, random_order_hint=False):
    rows = torch.from_numpy(rows_host.view(np.int32)).cuda()
    out = torch.empty((len(rows_host), reader.dim), dtype=torch.float32, device='cuda')
    ms = timer.launches(lambda: reader.rows_embedding_device(rows, out=out), launches)
    per_launch_median = ms[len(ms) // 2]
    median, minimum = per_launch_median, ms[0]
    if per_launch_median < 0.2:
        # kernels of less than 0.2 ms: bursts of launches between one pair of events each (Timer.bursts); median AND
        # minimum are burst averages (an event pair per launch adds 4-5 us, so the two methods must not be mixed)
        averages = timer.bursts(lambda: reader.rows_embedding_device(rows, out=out), max(launches, 50))
        median, minimum = averages[len(averages) // 2], averages[0]
    nbytes = algorithmic_bytes(library, reader, rows_host)
    parity = sampled_parity(path, rows_host, lambda picks: out[torch.from_numpy(picks).cuda()].cpu().numpy())
    info = reader.info(len(rows_host))
    result = {
        'workload': name,
        'what': what,
        'batch': len(rows_host),
        'kernel': info.get('kernel', ''),
        'kernel_ms': median,
        'kernel_ms_timing': 'median (kernel_min_ms: minimum) of per-launch HIP event pairs' if per_launch_median >= 0.2 else
                            'median (kernel_min_ms: minimum) of 5 bursts of back-to-back launches, one HIP event pair per burst, average per launch '
                            '(an event pair around every launch reads {:.4f} ms: 4-5 us of its own)'.format(per_launch_median),
        'kernel_min_ms': minimum,
        'embeddings_per_s': len(rows_host) / (median * 1e-3),
        'algorithmic_bytes': nbytes,
        'algorithmic_GBps': nbytes / (median * 1e-3) / 1e9,
        'frac': nbytes / (median * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        'parity': parity,
    }
    result['launch'] = {'waves_per_block': info.get('waves_per_block'), 'tiles_per_wavefront': info.get('tiles_per_wavefront')}
    if random_order_hint:
        # the same batch with the caller's hint MEMB_HIP_ROWS_IN_RANDOM_ORDER (Reader.rows_embedding_device(order='random')):
        # blocks of four wavefronts instead of the eight that key-order dumps like. The configuration's own figure above is
        # WITHOUT the hint -- what a caller who says nothing gets.
        out.zero_()
        hinted = timer.launches(lambda: reader.rows_embedding_device(rows, out=out, order='random'), launches)
        hinted_median = hinted[len(hinted) // 2]
        result['with_random_order_hint'] = {
            'what': 'order=\'random\' (MEMB_HIP_ROWS_IN_RANDOM_ORDER): blocks of four wavefronts', 'kernel_ms': hinted_median,
            'kernel_min_ms': hinted[0], 'frac': nbytes / (hinted_median * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            'parity': sampled_parity(path, rows_host, lambda picks: out[torch.from_numpy(picks).cuda()].cpu().numpy())}
    del rows, out
    return result


def rotating_batches(reader, timer, library, torch, np, repeated, batch=100000, sets=4):
    """configs[1] a second way. The figure above re-decodes ONE batch into ONE buffer: its 120 MB of output and ~16 MB of
    row regions stay in the 256 MB Infinity Cache from launch to launch, so its fraction of the HBM peak is cache-assisted
    (what a serving loop that reuses its buffers sees). Here `sets` different batches go round-robin into `sets` different
    output buffers (4 x 120 MB in flight > 256 MB): every launch's rows come from HBM and its output leaves for HBM."""
    count = len(reader)
    batches = []
    for k in range(sets):
        rng = np.random.default_rng(110 + k)
        rows = rng.integers(0, count, size=batch).astype(np.uint32)
        rows[rng.integers(0, batch, size=batch // 100)] = MISSING
        batches.append((rows, torch.from_numpy(rows.view(np.int32)).cuda(),
                        torch.empty((batch, reader.dim), dtype=torch.float32, device='cuda')))
    turn = [0]

    def call():
        _, ids, out = batches[turn[0] % sets]
        turn[0] += 1
        reader.rows_embedding_device(ids, out=out)

    averages = timer.bursts(call, 15 * sets)
    median = averages[len(averages) // 2]
    nbytes = sum(algorithmic_bytes(library, reader, rows) for rows, _, _ in batches) / sets
    # several of those batches in ONE launch (memb_hip_decode_batches_device / Reader.rows_embedding_device_many): launch gap,
    # prologue and tail once for all of them. Two groups of `sets` batches alternate, so that nothing is cached here either.
    more = []
    for k in range(sets):
        rng = np.random.default_rng(210 + k)
        rows = rng.integers(0, count, size=batch).astype(np.uint32)
        rows[rng.integers(0, batch, size=batch // 100)] = MISSING
        more.append((rows, torch.from_numpy(rows.view(np.int32)).cuda(),
                     torch.empty((batch, reader.dim), dtype=torch.float32, device='cuda')))
    groups = [[(ids, out) for _, ids, out in batches], [(ids, out) for _, ids, out in more]]

    def many():
        reader.rows_embedding_device_many(groups[turn[0] % 2])
        turn[0] += 1

    many_averages = timer.bursts(many, 30)
    many_median = many_averages[len(many_averages) // 2]
    many_bytes = sum(algorithmic_bytes(library, reader, rows) for rows, _, _ in batches + more) / 2
    import oracle
    checker = oracle.OracleReader(reader_path(reader), os.cpu_count() or 1) if reader_path(reader) else None
    many_parity = 'skipped'
    if checker is not None:
        torch.cuda.synchronize()
        picks = np.arange(0, batch, 37)
        many_parity = 'bit-exact ({} sampled rows of each of the {} batches)'.format(len(picks), 2 * sets)
        for rows, _, out in batches + more:
            if not np.array_equal(out[torch.from_numpy(picks).cuda()].cpu().numpy().view(np.uint32),
                                  checker.rows_embedding(np.ascontiguousarray(rows[picks])).view(np.uint32)):
                many_parity = 'MISMATCH'
    hbm = {
        'what': '{} different batches of {} rows round-robin into {} output buffers ({} MB in flight): nothing of a launch is still cached at its next turn'.format(
            sets, batch, sets, sets * batch * reader.dim * 4 // 1000000),
        'kernel_ms': median, 'kernel_min_ms': averages[0],
        'kernel_ms_timing': 'median / minimum of 5 bursts of {} launches, one HIP event pair per burst'.format(15 * sets),
        'algorithmic_GBps': nbytes / (median * 1e-3) / 1e9,
        'frac': nbytes / (median * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        'against_repeated_buffer': median / repeated['kernel_ms'],
    }
    return {
        'hbm': hbm,
        'batches_in_one_launch': {
            'what': '{} such batches in ONE launch (memb_hip_decode_batches_device: tiles numbered through, one prologue and one tail), two groups of '
                    '{} batches and buffers alternating ({} MB in flight)'.format(sets, sets, 2 * sets * batch * reader.dim * 4 // 1000000),
            'kernel': 'decode_trained_batches', 'batches': sets, 'launch_ms': many_median, 'launch_min_ms': many_averages[0],
            'ms_per_batch': many_median / sets,
            'algorithmic_GBps': many_bytes / (many_median * 1e-3) / 1e9,
            'frac': many_bytes / (many_median * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            'against_one_batch_per_launch': (many_median / sets) / median,
            'parity': many_parity,
        },
    }


_READER_PATHS = {}


def reader_path(reader):
    return _READER_PATHS.get(id(reader))


def primary_is_hbm(entry):
    """configs[1]: the figure of the configuration is the one in which every byte comes from and goes to HBM (four
    batches round-robin into four buffers); the cache-assisted one (one batch re-decoded into one buffer) moves to
    `repeated_buffer`."""
    hbm = entry.pop('hbm')
    entry['repeated_buffer'] = {
        'what': 'cache-assisted: ONE batch decoded again and again into ONE buffer (output and row regions of a launch still in the 256 MB Infinity Cache at the next)',
        'kernel_ms': entry['kernel_ms'], 'kernel_min_ms': entry['kernel_min_ms'], 'kernel_ms_timing': entry['kernel_ms_timing'],
        'algorithmic_GBps': entry['algorithmic_GBps'], 'frac': entry['frac'], 'embeddings_per_s': entry['embeddings_per_s'],
    }
    entry['frac_is'] = 'HBM regime: ' + hbm['what']
    for key in ('kernel_ms', 'kernel_min_ms', 'kernel_ms_timing', 'algorithmic_GBps', 'frac'):
        entry[key] = hbm[key]
    entry['embeddings_per_s'] = entry['batch'] / (hbm['kernel_ms'] * 1e-3)
    entry['against_repeated_buffer'] = hbm['against_repeated_buffer']


def measure_union(reader_a, path_a, reader_b, path_b, timer, library, torch, np, batch=500000, launches=15):
    """BASELINE.json configs[4]: ReadersUnion 'concatenate' of two 4-bit models, 500 000 words, (n, 600) output;
    a quarter of the words is missing from each model (so about half of the words are known to both)."""
    import oracle
    from memb_amd import _memb
    rng = np.random.default_rng(17)
    rows_a = rng.integers(0, len(reader_a), size=batch).astype(np.uint32)
    rows_a[rng.random(batch) < 0.25] = MISSING
    rows_b = rng.integers(0, len(reader_b), size=batch).astype(np.uint32)
    rows_b[rng.random(batch) < 0.25] = MISSING
    ids = [torch.from_numpy(rows_a.view(np.int32)).cuda(), torch.from_numpy(rows_b.view(np.int32)).cuda()]
    width = reader_a.dim + reader_b.dim
    merged = torch.empty((batch, width), dtype=torch.float32, device='cuda')
    stream = torch.cuda.current_stream().cuda_stream

    def fused():
        return _memb.union_rows_to_device(
            [reader_a._impl, reader_b._impl], [ids[0].data_ptr(), ids[1].data_ptr()], [0, reader_a.dim], batch,
            merged.data_ptr(), merged.stride(0), stream, False)

    def per_reader():   # models of different key formats cannot share the kernel: one launch per column block
        reader_a.rows_embedding_device(ids[0], out=merged, col_off=0)
        reader_b.rows_embedding_device(ids[1], out=merged, col_off=reader_a.dim)

    one_launch = bool(fused())
    ms = timer.launches(fused if one_launch else per_reader, launches)
    median = ms[len(ms) // 2]
    nbytes = algorithmic_bytes(library, reader_a, rows_a) + algorithmic_bytes(library, reader_b, rows_b) - 4 * batch
    picks = np.sort(rng.choice(batch, size=20000, replace=False))
    cores = os.cpu_count() or 1
    expected = np.concatenate([
        oracle.OracleReader(path_a, cores).rows_embedding(np.ascontiguousarray(rows_a[picks])),
        oracle.OracleReader(path_b, cores).rows_embedding(np.ascontiguousarray(rows_b[picks]))], axis=-1)
    got = merged[torch.from_numpy(picks).cuda()].cpu().numpy()
    parity = 'bit-exact (20000 sampled rows)' if np.array_equal(got.view(np.uint32), expected.view(np.uint32)) else 'MISMATCH'
    # the configuration as the reference states it -- WORDS in: ReadersUnion.batch_embedding_device packs the words once,
    # both readers resolve them on the device (resolve_words) and the fused kernel decodes from the row ids in HBM
    from_words = None
    if os.environ.get('MEMB_BENCH_REHEARSAL') != 'cpu':
        import memb_amd
        keys_a, keys_b = reader_a.keys(), reader_b.keys()
        word_rng = np.random.default_rng(19)
        words = [keys_a[i] for i in word_rng.integers(0, len(keys_a), size=batch // 2)] + \
                [keys_b[i] for i in word_rng.integers(0, len(keys_b), size=batch - batch // 2)]
        for i in range(0, batch, 8):
            words[i] = words[i] + '~'   # an eighth of the words is in neither model
        union = memb_amd.ReadersUnion([reader_a, reader_b], 'concatenate')
        union.batch_embedding_device(words)
        torch.cuda.synchronize()
        best = float('inf')
        for _ in range(5):
            start = time.perf_counter()
            result = union.batch_embedding_device(words)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - start)
        word_picks = np.sort(word_rng.choice(batch, size=5000, replace=False))
        sample = [words[i] for i in word_picks]
        want = np.concatenate([oracle.OracleReader(path_a, cores).batch_embedding(sample),
                               oracle.OracleReader(path_b, cores).batch_embedding(sample)], axis=-1)
        have = result[torch.from_numpy(word_picks).cuda()].cpu().numpy()
        from_words = {
            'what': 'ReadersUnion([glove, fasttext], concatenate).batch_embedding_device(list of {} str): strings -> pinned memory once, resolve_words per '
                    'reader, the fused kernel; call to merged rows in HBM incl. synchronize, best of 5'.format(batch),
            'ms': best * 1e3, 'words_per_s': batch / best,
            'parity': 'bit-exact (5000 sampled words against the CPU checker: search + decode)' if np.array_equal(have.view(np.uint32), want.view(np.uint32)) else 'MISMATCH',
        }
        del result, union, words, keys_a, keys_b
    return {
        'from_words': from_words,
        'workload': 'union-concat-glove4bit+fasttext4bit-500k (BASELINE.json configs[4])',
        'what': 'ReadersUnion concatenate, two 4-bit models, 500 000 words, 25 % of them missing per model, (n, 600) fp32 output, ' +
                ('one launch of the fused kernel (named in `kernel`, as the library reports it)' if one_launch else 'one launch per reader (key formats differ)'),
        'batch': batch,
        'kernel': (reader_a.info().get('union_kernel') or 'fused union kernel') if one_launch else 'decode_trained x 2',
        'kernel_ms': median,
        'kernel_min_ms': ms[0],
        'embeddings_per_s': batch / (median * 1e-3),
        'algorithmic_bytes': nbytes,
        'algorithmic_GBps': nbytes / (median * 1e-3) / 1e9,
        'frac': nbytes / (median * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        'parity': parity,
    }


def recorded_traffic(results, names):
    """HBM bytes per launch from the PMC counters (profiles/hbm_traffic.json: rocprofv3 passes over each workload, tools/perf/prof.sh)
    next to the algorithmic bytes of every configuration that has them; static, like roofline.traffic of the main line."""
    path = os.path.join(REPO, 'profiles', 'hbm_traffic.json')
    if not os.path.exists(path):
        return
    with open(path) as f:
        recorded = json.load(f)
    for entry in results:
        for prefix, key in names:
            if entry['workload'].startswith(prefix) and recorded.get(key) is not None and 'algorithmic_bytes' in entry:
                entry['traffic'] = recorded[key]
                entry['traffic_over_algorithmic'] = recorded[key] / entry['algorithmic_bytes']
                entry['traffic_source'] = 'static: profiles/hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over --workload {})'.format(key)


def all_configs(args, memb_amd, synthetic, headline, timer, library, torch, np, sizes, ceilings=None):
    """Every BASELINE.json configuration, measured on cuda:0 outside the timed region."""
    glove, fasttext = sizes
    reader4, path4 = headline
    results = []
    build_seconds = 0.0

    # configs[0]: 1k-word uniform 8-bit (the reference's CPU-runnable plumbing case), here through the HIP path
    path, spent = synthetic.cached_model(1000, 300, 'uniform', 8)
    build_seconds += spent
    reader = memb_amd.Reader(path, device=0)
    rows = np.concatenate([np.arange(1000, dtype=np.uint32), np.full(100, MISSING, dtype=np.uint32)])
    results.append(measure_config(
        'uniform-8bit-1k (BASELINE.json configs[0])', '1 000-word uniform 8-bit model, all keys + 10 % misses; 7 us of kernel: launch latency, not bandwidth',
        reader, path, rows, timer, library, torch, np))
    del reader

    rows = batch_rows(len(reader4), 100000, np)
    results.append(measure_config(
        'glove840b-300d-4bit-100k (BASELINE.json configs[1])', '100 000 uniformly random rows of the 2.2 M-word 4-bit model, 1 % misses',
        reader4, path4, rows, timer, library, torch, np, launches=30))
    _READER_PATHS[id(reader4)] = path4
    results[-1].update(rotating_batches(reader4, timer, library, torch, np, results[-1]))
    primary_is_hbm(results[-1])
    # device-resident latency of small batches of the same model: the kernel is chosen by batch size (one tile per
    # wavefront / decode_records_persistent / large-batch kernel), each timed as a burst of back-to-back launches
    small = []
    for count in (1000, 10000, 100000, 500000):
        picks = batch_rows(len(reader4), count, np)
        ids = torch.from_numpy(picks.view(np.int32)).cuda()
        target = torch.empty((count, reader4.dim), dtype=torch.float32, device='cuda')
        # (up to 10 000 rows the launches are replayed from ONE HIP graph: a 5 us kernel against 5-8 us per Python call on some hosts)
        if count <= 10000:
            ms, launched = timer.graph_burst(lambda: reader4.rows_embedding_device(ids, out=target), 100)
        else:
            ms, launched = timer.burst(lambda: reader4.rows_embedding_device(ids, out=target), 100 if count <= 100000 else 30), 'eager'
        nbytes = algorithmic_bytes(library, reader4, picks)
        small.append({'batch': count, 'kernel': reader4.info(count)['kernel'], 'lanes_per_word': reader4.info(count)['lanes_per_word'], 'us_per_launch': ms * 1e3,
                      'launched': launched,
                      'embeddings_per_s': count / (ms * 1e-3), 'frac': nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS})
        del ids, target
    results[-1]['small_batches_of_the_same_model'] = small

    # the headline's batch in RANDOM order (token streams are: reference src/reader.cpp:49-57 takes words in any order): every
    # row region is then two 128-byte lines of its own where a key-order dump reads 1.25 lines per row
    shuffled4 = np.random.default_rng(77).permutation(len(reader4)).astype(np.uint32)
    results.append(measure_config(
        'glove840b-300d-4bit-fullvocab-shuffled (the headline batch in random order)',
        'every row of the 2.2 M-word 4-bit model once, in shuffled order', reader4, path4, shuffled4, timer, library, torch, np,
        random_order_hint=True))
    if ceilings and ceilings.get('tile_fill_random_records'):
        results[-1]['pattern_ceiling'] = {'tile_fill_random_records_ms': ceilings['tile_fill_random_records']['ms'],
                                          'kernel_over_ceiling': results[-1]['kernel_ms'] / ceilings['tile_fill_random_records']['ms']}
    del shuffled4

    path, spent = synthetic.cached_model(fasttext, 300, 'trained', 6)
    build_seconds += spent
    reader = memb_amd.Reader(path, device=0)
    results.append(measure_config(
        'fasttext2m-300d-6bit-fullvocab (BASELINE.json configs[2])', 'full dump of a 2.0 M-word 6-bit model (byte keys, codes up to 10 bits)',
        reader, path, np.arange(len(reader), dtype=np.uint32), timer, library, torch, np))
    results.append(measure_config(
        'fasttext2m-300d-6bit-fullvocab-shuffled (configs[2]\'s batch in random order)',
        'every row of the 2.0 M-word 6-bit model once, in shuffled order', reader, path,
        np.random.default_rng(78).permutation(len(reader)).astype(np.uint32), timer, library, torch, np, random_order_hint=True))
    del reader

    path, spent = synthetic.cached_model(glove, 300, 'trained', 2)
    build_seconds += spent
    reader = memb_amd.Reader(path, device=0)
    results.append(measure_config(
        'glove840b-300d-2bit-fullvocab (BASELINE.json configs[3], one GPU\'s view: the whole dump)', 'full dump of the 2.2 M-word 2-bit model on one GPU',
        reader, path, np.arange(len(reader), dtype=np.uint32), timer, library, torch, np))
    del reader

    path, spent = synthetic.cached_model(fasttext, 300, 'trained', 4, seed=4321)
    build_seconds += spent
    reader = memb_amd.Reader(path, device=0)
    results.append(measure_union(reader4, path4, reader, path, timer, library, torch, np,
                                 batch=min(500000, len(reader4))))
    del reader

    # The headline's model is the decoder's best case: i.i.d. Gaussian weights whose 4-bit code tops out at exactly 8 bits,
    # the limit of the nibble-key path (hip_trained_kernels.h, FAST). Two full dumps off that path: a seed whose code has a
    # 9-bit word (byte keys: 4-byte table entries, one symbol per byte of the tile), and heavier-tailed Student-t(5) * 0.3
    # weights, where k-means (reference src/kmeans.cpp:53-60) keeps fewer centroids and the streams are shorter.
    for seed, distribution, label, what in (
            (OFF_PATH_SEED, 'normal', 'glove840b-300d-4bit-fullvocab-bytekeys', 'full dump, N(0, 0.4^2) seed {}: a code longer than 8 bits'.format(OFF_PATH_SEED)),
            (1234, 'student', 'glove840b-300d-4bit-fullvocab-student-t', 'full dump, Student-t(5) * 0.3 weights')):
        path, spent = synthetic.cached_model(glove, 300, 'trained', 4, seed=seed, distribution=distribution)
        build_seconds += spent
        reader = memb_amd.Reader(path, device=0)
        entry = measure_config(label + ' (off the headline\'s happy path)', what, reader, path,
                               np.arange(len(reader), dtype=np.uint32), timer, library, torch, np)
        facts = reader.info()
        entry['max_code_bits'] = facts['max_code_bits']
        entry['key_format'] = 'nibble keys (<= 16 centroids, codes <= 8 bits)' if entry['kernel'].rstrip('>').split(',')[2].strip() == 'true' else 'byte keys'
        entry['row_bytes'] = facts['row_bytes']
        results.append(entry)
        del reader

    count = min(500000, glove)
    path, spent = synthetic.cached_model(count, 300, 'uniform', 8)
    build_seconds += spent
    reader = memb_amd.Reader(path, device=0)
    results.append(measure_config(
        'uniform-8bit-500k', 'full dump of a 500 000-word uniform 8-bit model (bit-exact dequantisation, four IEEE fp32 operations per weight)',
        reader, path, np.arange(len(reader), dtype=np.uint32), timer, library, torch, np))
    del reader
    if not args.small:
        recorded_traffic(results, (('glove840b-300d-4bit-100k', 'glove840b-300d-4bit-100k'), ('fasttext2m-300d-6bit-fullvocab', 'fasttext2m-300d-6bit-fullvocab'),
                                   ('glove840b-300d-2bit-fullvocab (', 'glove840b-300d-2bit-fullvocab'), ('union-concat', 'union-concat-500k'),
                                   ('uniform-8bit-500k', 'uniform-8bit-500k')))
    return results, build_seconds


def special_workload(name, args, memb_amd, synthetic, library, torch, np, glove, fasttext):
    """The two configurations that are not one trained model, as the timed step of the main line (so that
    `rocprofv3 ... -- python3 bench.py --workload <name> --no-configs` profiles exactly that kernel):
    returns step(), the output tensor, batch size, algorithmic bytes, a parity callable, a description."""
    import oracle
    from memb_amd import _memb
    cores = os.cpu_count() or 1
    build_seconds = 0.0
    if name == 'uniform-8bit-500k':
        count = min(500000, glove)
        path, spent = synthetic.cached_model(count, 300, 'uniform', 8)
        build_seconds += spent
        reader = memb_amd.Reader(path, device=0)
        info = reader.info()
        rows_host = np.arange(len(reader), dtype=np.uint32)
        rows = torch.from_numpy(rows_host.view(np.int32)).cuda()
        out = torch.empty((len(rows_host), reader.dim), dtype=torch.float32, device='cuda')
        return {
            'step': lambda: reader.rows_embedding_device(rows, out=out), 'out': out, 'n': len(rows_host),
            'nbytes': algorithmic_bytes(library, reader, rows_host), 'kernel': info['kernel'], 'info': info, 'keep': (reader, rows),
            'parity': lambda: sampled_parity(path, rows_host, lambda picks: out[torch.from_numpy(picks).cuda()].cpu().numpy()),
            'config': {'vocabulary': len(reader), 'dim': reader.dim, 'storage': 'uniform', 'bits_per_weight': 8,
                       'batch': 'keys() full dump'},
            'build_seconds': build_seconds,
        }
    path_a, spent = synthetic.cached_model(glove, 300, 'trained', 4)
    build_seconds += spent
    path_b, spent = synthetic.cached_model(fasttext, 300, 'trained', 4, seed=4321)
    build_seconds += spent
    reader_a = memb_amd.Reader(path_a, device=0)
    reader_b = memb_amd.Reader(path_b, device=0)
    info = reader_a.info()
    reader_b.info()
    batch = min(500000, len(reader_a))
    rng = np.random.default_rng(17)   # the batch of measure_union
    rows_a = rng.integers(0, len(reader_a), size=batch).astype(np.uint32)
    rows_a[rng.random(batch) < 0.25] = MISSING
    rows_b = rng.integers(0, len(reader_b), size=batch).astype(np.uint32)
    rows_b[rng.random(batch) < 0.25] = MISSING
    ids = [torch.from_numpy(rows_a.view(np.int32)).cuda(), torch.from_numpy(rows_b.view(np.int32)).cuda()]
    merged = torch.empty((batch, reader_a.dim + reader_b.dim), dtype=torch.float32, device='cuda')

    def step():
        done = _memb.union_rows_to_device(
            [reader_a._impl, reader_b._impl], [ids[0].data_ptr(), ids[1].data_ptr()], [0, reader_a.dim], batch,
            merged.data_ptr(), merged.stride(0), torch.cuda.current_stream().cuda_stream, False)
        if not done:
            raise SystemExit('the two models cannot share decode_trained_union')

    def parity():
        picks = np.sort(np.random.default_rng(5).choice(batch, size=min(20000, batch), replace=False))
        expected = np.concatenate([
            oracle.OracleReader(path_a, cores).rows_embedding(np.ascontiguousarray(rows_a[picks])),
            oracle.OracleReader(path_b, cores).rows_embedding(np.ascontiguousarray(rows_b[picks]))], axis=-1)
        got = merged[torch.from_numpy(picks).cuda()].cpu().numpy()
        return 'bit-exact ({} sampled rows)'.format(len(picks)) if np.array_equal(got.view(np.uint32), expected.view(np.uint32)) else 'MISMATCH'

    return {
        'step': step, 'out': merged, 'n': batch,
        'nbytes': algorithmic_bytes(library, reader_a, rows_a) + algorithmic_bytes(library, reader_b, rows_b) - 4 * batch,
        'kernel': 'decode_union_split', 'kernel_of': lambda: reader_a.info().get('union_kernel') or 'decode_union_split',
        'info': info, 'keep': (reader_a, reader_b, ids), 'parity': parity,
        'config': {'vocabulary': [len(reader_a), len(reader_b)], 'dim': reader_a.dim + reader_b.dim, 'storage': 'trained + trained',
                   'bits_per_weight': 4, 'batch': '500 000 random words, 25 % missing per model, ReadersUnion concatenate in one launch'},
        'build_seconds': build_seconds,
    }


# --------------------------------------------------------------------------------------------
# strong scaling: ONE dump split over the ranks
# --------------------------------------------------------------------------------------------

def strong_scaling(args, memb_amd, synthetic, rank, world_size, local_rank, dist, torch, np, library, words, bits, name):
    from memb_amd.sharding import shard_range
    distributed = world_size > 1
    build_seconds = 0.0
    if rank == 0:
        path, build_seconds = synthetic.cached_model(words, 300, 'trained', bits)
    if distributed:
        dist.barrier()
    path, _ = synthetic.cached_model(words, 300, 'trained', bits)
    reader, info, _ = open_reader(memb_amd, path, local_rank)
    count = len(reader)
    start, stop = shard_range(count, rank, world_size)   # the reference's thread split, src/reader.cpp:65-79
    mine = np.arange(start, stop, dtype=np.uint32)
    rows = torch.from_numpy(mine.view(np.int32)).cuda()
    out = torch.empty((len(mine), reader.dim), dtype=torch.float32, device='cuda')
    host = torch.empty((len(mine), reader.dim), dtype=torch.float32, pin_memory=True)
    steps = args.steps

    def timed(call):
        call()
        torch.cuda.synchronize()
        probe = time.perf_counter()
        call()
        torch.cuda.synchronize()
        one = max(time.perf_counter() - probe, 1e-6)
        for _ in range(max(3, min(2000, int(0.02 / one) + 1))):   # ~20 ms without a gap: the power state settles
            call()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        begin = time.perf_counter()
        for _ in range(steps):
            call()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - begin
        if distributed:
            slowest = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
            dist.all_reduce(slowest, op=dist.ReduceOp.MAX)
            elapsed = float(slowest.item())
        return elapsed

    def kernel_only():
        reader.rows_embedding_device(rows, out=out)

    def with_d2h():
        reader.rows_embedding_device(rows, out=out)
        host.copy_(out, non_blocking=True)

    elapsed_kernel = timed(kernel_only)
    kernel_ms = Timer(torch).launches(kernel_only, steps)
    elapsed_d2h = timed(with_d2h)

    # the product's own host-side gather: every rank decodes its slice straight into its rows of a host
    # matrix (memb_hip_decode_rows: centroid indices over PCIe, expanded by host threads)
    host_rows = np.zeros((len(mine), reader.dim), dtype=np.float32)

    def host_gather():
        reader.rows_embedding_into(mine, host_rows)

    elapsed_host = timed(host_gather)
    host_parity = sampled_parity(path, mine, lambda picks: host_rows[picks], sample=5000)
    nbytes = algorithmic_bytes(library, reader, mine)
    parity = sampled_parity(path, mine, lambda picks: host[torch.from_numpy(picks)].numpy(), sample=5000)
    mine_summary = {
        'rank': rank, 'device': local_rank, 'rows': [int(start), int(stop)],
        'kernel_avg_ms': sum(kernel_ms) / len(kernel_ms), 'kernel_min_ms': kernel_ms[0],
        'algorithmic_GBps': nbytes / (sum(kernel_ms) / len(kernel_ms) * 1e-3) / 1e9, 'parity': parity,
    }
    if distributed:
        gathered = [None] * world_size
        dist.all_gather_object(gathered, mine_summary)
    else:
        gathered = [mine_summary]
    del reader, rows, out, host
    return {
        'workload': name + ' (BASELINE.json configs[3])',
        'what': 'ONE {}-word {}-bit dump split over {} rank(s), rank g decodes rows [g*ceil(n/G), (g+1)*ceil(n/G)) into its own device buffer; no collective'.format(count, bits, world_size),
        'scaling': 'strong',
        'n_gpus': world_size,
        'ranks_seen': len(gathered),
        'steps': steps,
        'kernel_only': {'value': count * steps / elapsed_kernel, 'unit': 'embeddings/s', 'ms_per_step': elapsed_kernel / steps * 1e3},
        'with_d2h': {'value': count * steps / elapsed_d2h, 'unit': 'embeddings/s', 'ms_per_step': elapsed_d2h / steps * 1e3,
                     'note': 'each rank also copies its slice of the fp32 result into its own pinned host buffer (PCIe-bound); the disjoint slices of those buffers are the host-side gather'},
        'host_gather': {'value': count * steps / elapsed_host, 'unit': 'embeddings/s', 'ms_per_step': elapsed_host / steps * 1e3,
                        'parity_rank0': host_parity,
                        'note': 'Reader.rows_embedding_into per rank: its slice decoded into its rows of a host matrix through the product\'s host-buffer path (centroid indices over PCIe, host threads expand them)'},
        'per_rank': gathered,
    }, build_seconds


# --------------------------------------------------------------------------------------------

def main():
    args = parse_args()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world_size = int(os.environ.get('WORLD_SIZE', '1'))
    args.gpus = world_size

    import numpy as np
    import torch
    import torch.distributed as dist

    import __graft_entry__
    if rank == 0 and os.environ.get('MEMB_BENCH_PREBUILT') != '1':
        __graft_entry__.build()
    distributed = world_size > 1
    # Rehearsal of the multi-rank plumbing on a one-GPU box: MEMB_BENCH_REHEARSAL=1 puts every rank on
    # cuda:0 and uses gloo (RCCL refuses two ranks on one device). Never set by the driver.
    cpu_rehearsal = os.environ.get('MEMB_BENCH_REHEARSAL') == 'cpu'   # no GPU at all: install_host_stand_ins
    rehearsal = distributed and os.environ.get('MEMB_BENCH_REHEARSAL') in ('1', 'cpu')
    if cpu_rehearsal:
        import memb_amd
        install_host_stand_ins(torch, memb_amd)
        args.host_writer = args.no_ceilings = True
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # The process group carries barriers, one MAX all-reduce of the elapsed time and the gather of the per-rank
        # summaries -- nothing of the data path (tests/test_bench_contract.py greps this file for any other collective).
        if rehearsal:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        dist.barrier()
    import memb_amd
    from memb_amd import synthetic
    from memb_amd.sharding import shard_range

    strong_main = args.scaling == 'strong'
    workload = args.workload
    if strong_main and workload == 'glove840b-300d-4bit-fullvocab':
        workload = STRONG_WORKLOAD
    words, bits, batch = WORKLOADS[workload]
    glove, fasttext = GLOVE_WORDS, FASTTEXT_WORDS
    if args.small:
        words, glove, fasttext = min(words, 50000), 50000, 49999
    os.environ['MEMB_BENCH_CACHE'] = args.cache_dir
    # the synthetic models are written through the device writer (quantisation, histogram and bit packing on
    # this rank's GPU; byte-identical files: tests/test_gpu_writer.py) unless asked otherwise
    if args.host_writer:
        os.environ.pop('MEMB_SYNTH_DEVICE', None)
    else:
        os.environ['MEMB_SYNTH_DEVICE'] = str(local_rank)
    build_seconds = 0.0
    if rank == 0:
        # (count, dim, storage, bits, seed) of everything this run opens, written up front and side by side
        needed = [(words, 300, 'trained', bits, 1234)] if workload not in SPECIAL_WORKLOADS else []
        if workload == 'union-concat-500k':
            needed += [(glove, 300, 'trained', 4, 1234), (fasttext, 300, 'trained', 4, 4321)]
        if workload == 'uniform-8bit-500k':
            needed += [(min(500000, glove), 300, 'uniform', 8, 1234)]
        if not args.no_configs and workload not in SPECIAL_WORKLOADS:
            if world_size == 1:
                needed += [(glove, 300, 'trained', 4, 1234), (fasttext, 300, 'trained', 6, 1234), (glove, 300, 'trained', 2, 1234),
                           (fasttext, 300, 'trained', 4, 4321), (min(500000, glove), 300, 'uniform', 8, 1234), (1000, 300, 'uniform', 8, 1234),
                           (glove, 300, 'trained', 4, OFF_PATH_SEED), (glove, 300, 'trained', 4, 1234, 'student')]
            else:
                needed += [(glove, 300, 'trained', 2, 1234)]
        build_seconds = prebuild_models(synthetic, list(dict.fromkeys(needed)))
    library = ctypes.CDLL(memb_amd.HIP_LIBRARY_PATH)
    timer = Timer(torch)
    special = None
    if workload in SPECIAL_WORKLOADS:
        if distributed or strong_main:
            raise SystemExit('--workload {} is a one-GPU profiling run'.format(workload))
        args.no_configs = True
        args.no_cpu_baseline = True
        special = special_workload(workload, args, memb_amd, synthetic, library, torch, np, glove, fasttext)
        build_seconds += special['build_seconds']
        info, open_seconds, out, n, nbytes, step = special['info'], 0.0, special['out'], special['n'], special['nbytes'], special['step']
        dim, count, rows_host, reader, path = out.shape[1], special['n'], None, None, None
    else:
        if distributed:
            dist.barrier()   # (rank 0 has written the models: prebuild_models above)
        path, _ = synthetic.cached_model(words, 300, 'trained', bits)

        reader, info, open_seconds = open_reader(memb_amd, path, local_rank, batch or 0)
        dim = reader.dim
        count = len(reader)

        rows_all = batch_rows(count, batch, np)
        if strong_main:
            start, stop = shard_range(len(rows_all), rank, world_size)
            rows_host = np.ascontiguousarray(rows_all[start:stop])
        else:
            rows_host = rows_all
        n = len(rows_host)
        nbytes = algorithmic_bytes(library, reader, rows_host)
        rows = torch.from_numpy(rows_host.view(np.int32)).cuda()
        out = torch.empty((n, dim), dtype=torch.float32, device='cuda')
        if batch is not None and not strong_main and n * dim * 4 < 256 * 1024 * 1024:
            # A batch whose output fits the caches (100 000 rows = 120 MB against 256 MB of Infinity Cache): the same batch into
            # the same buffer step after step would be timed -- and profiled -- out of the caches. FOUR different batches
            # round-robin into four buffers instead, as `configs[1]`'s own figure is taken (rotating_batches); the first is the
            # configured batch and the one whose output is checked.
            rotation = [(rows, out)]
            rotating_bytes = [nbytes]
            for seed in (12, 13, 14):
                rng = np.random.default_rng(seed)
                other = rng.integers(0, count, size=n).astype(np.uint32)
                other[rng.integers(0, n, size=n // 100)] = MISSING
                rotating_bytes.append(algorithmic_bytes(library, reader, other))
                rotation.append((torch.from_numpy(other.view(np.int32)).cuda(), torch.empty((n, dim), dtype=torch.float32, device='cuda')))
            nbytes = sum(rotating_bytes) // len(rotating_bytes)
            turn = [0]

            def step():
                ids, target = rotation[turn[0] % len(rotation)]
                turn[0] += 1
                reader.rows_embedding_device(ids, out=target)
        else:
            rotation = None

            def step():
                reader.rows_embedding_device(rows, out=out)

    # Events first, so that nothing but launches lies between the phases below.
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    stops = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    fills = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(FILL_LAUNCHES)]
    torch.cuda.synchronize()

    # (1) The box's yardstick: torch's fill_ of the very same output buffer, FILL_LAUNCHES times (~20 ms),
    # enqueued directly in front of the warmup. It doubles as the run-in of the GPU's power state: after
    # any idle gap of a few milliseconds this part runs launches 3..25 of a burst ~10 % slower than
    # launch 26 onwards (tools/perf/ramp.py: 0.65 ms against 0.59 ms, whether the gap was 0, 3 or 10 s),
    # and with W = 5, K = 20 the timed region would sit exactly inside that transient.
    for begin, end in fills:
        begin.record()
        out.fill_(0.0)
        end.record()
    # (2) W untimed warmup steps
    for _ in range(args.warmup):
        step()
    # (3) the timed region: exactly `steps` steps between barrier + synchronize on both sides; per-launch
    # kernel durations from HIP events on the stream the kernel runs on
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    wall_start = time.perf_counter()
    for i in range(args.steps):
        starts[i].record()
        step()
        stops[i].record()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - wall_start
    if distributed:
        slowest = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(slowest, op=dist.ReduceOp.MAX)
        elapsed = float(slowest.item())
    kernel_ms = sorted(starts[i].elapsed_time(stops[i]) for i in range(args.steps))
    if special is None:
        info = reader.info(n)   # (kernel and launch geometry are chosen by batch size: a static rule, memb_hip.hip planTrained)
    kernel_avg_ms = sum(kernel_ms) / len(kernel_ms)
    kernel_min_ms, kernel_median_ms = kernel_ms[0], kernel_ms[len(kernel_ms) // 2]
    kernel_timing = 'HIP event pair around every launch of the timed region: average (kernel_min_ms, kernel_median_ms: of the same list)'
    if kernel_avg_ms < 0.2:
        # an event pair per launch costs 4-5 us of its own: short kernels are quoted from bursts of K launches
        # between one pair of events each, right after the timed region (Timer.bursts) -- average, minimum and
        # median all from those bursts
        per_launch_avg = kernel_avg_ms
        averages = timer.bursts(step, args.steps)
        kernel_avg_ms, kernel_min_ms, kernel_median_ms = sum(averages) / len(averages), averages[0], averages[len(averages) // 2]
        kernel_timing = ('5 bursts of {} back-to-back launches, one HIP event pair per burst, after the timed region: average per launch over the bursts '
                         '(kernel_min_ms / kernel_median_ms: fastest / median burst; event pairs around every launch of the timed region averaged {:.4f} ms)').format(args.steps, per_launch_avg)
    fill_ms = sorted(begin.elapsed_time(end) for begin, end in fills[FILL_LAUNCHES // 2:])   # the settled half
    rank_summary = {'rank': rank, 'device': local_rank, 'batch': n, 'kernel_avg_ms': kernel_avg_ms, 'kernel_min_ms': kernel_min_ms,
                    'reader_open_s': open_seconds, 'device_bytes': info.get('device_bytes'), 'word_index_bytes': info.get('word_index_bytes')}
    if distributed:
        per_rank = [None] * world_size
        dist.all_gather_object(per_rank, rank_summary)
    else:
        per_rank = [rank_summary]

    # BASELINE.json configs[3] split over the ranks (every rank takes part; outside the timed region)
    strong = None
    if not strong_main and not args.no_configs and distributed:
        strong, spent = strong_scaling(
            args, memb_amd, synthetic, rank, world_size, local_rank, dist, torch, np, library,
            glove, 2, STRONG_WORKLOAD)
        build_seconds += spent

    if rank != 0:
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return

    # parity of the timed output against the CPU checker, and the CPU baseline
    baseline = None
    parity = 'skipped'
    # (the CPU legs run at N = 1 only: with more ranks the others would wait at the final barrier)
    cpu_legs = not args.no_cpu_baseline and world_size == 1
    if special is None and rotation is not None:
        reader.rows_embedding_device(rows, out=out)   # (the configured batch is the one that is checked)
        torch.cuda.synchronize()
    if cpu_legs:
        baseline, expected = cpu_baseline(path, rows_host, dim)
        got = out.cpu().numpy()
        parity = 'bit-exact' if np.array_equal(got.view(np.uint32), expected.view(np.uint32)) else 'MISMATCH'
        del expected, got
    elif special is not None:
        parity = special['parity']()
    elif not args.no_cpu_baseline:
        # N > 1: no timed CPU leg, but rank 0's output is still checked on a sample of its batch
        parity = sampled_parity(path, rows_host, lambda picks: out[torch.from_numpy(picks).cuda()].cpu().numpy())

    # Not part of `value`: what a caller of the reference's API sees (words in, numpy out; word search,
    # PCIe and host memory included), next to the restated CPU Reader on the same words and host cores.
    host_api = None
    word_search = None
    if cpu_legs:
        host_api = host_api_timings(reader, path, rows_host)
        if not cpu_rehearsal:
            word_search = word_search_timings(reader, path, torch, np)

    # the box's own ceilings for this memory pattern (no decoder): same buffer, same run-in
    ceilings = None
    if special is None and batch is None and not args.no_ceilings:
        merged = torch.empty((min(500000, n), 2 * dim), dtype=torch.float32, device='cuda') if dim == 300 else None
        ceilings = box_ceilings(torch, timer, out, n, union=(merged, merged.shape[0]) if merged is not None else None) if dim == 300 else None
        del merged
        if ceilings:
            # the timed kernel against the FASTEST pattern with the same reads this box has shown (a key-order dump: sequential records)
            candidates = {key: ceilings[key]['ms'] for key in ('tile_fill_sequential_records', 'persistent_tile_fill_sequential_records',
                                                               'two_tiles_sequential_records') if key in ceilings}
            if candidates:
                best = min(candidates, key=candidates.get)
                ceilings['kernel_against_the_fastest_pattern'] = {
                    'pattern': best, 'pattern_ms': candidates[best], 'kernel_avg_ms': kernel_avg_ms, 'kernel_over_pattern': kernel_avg_ms / candidates[best]}

    configs = None
    if world_size == 1 and not args.no_configs:
        del out, rows
        configs, spent = all_configs(args, memb_amd, synthetic, (reader, path), timer, library, torch, np, (glove, fasttext), ceilings)
        build_seconds += spent

    achieved_gbps = nbytes / (kernel_avg_ms * 1e-3) / 1e9
    traffic = None
    traffic_source = None
    traffic_file = os.path.join(REPO, 'profiles', 'hbm_traffic.json')
    recorded = {}
    if os.path.exists(traffic_file):
        with open(traffic_file) as f:
            recorded = json.load(f)
    why_not_live = None
    if world_size == 1 and not strong_main and not args.no_live_traffic and not args.small and not cpu_rehearsal:
        # the counters of THIS run (two child processes under rocprofv3; the parent's timed region is long over)
        kernel_name = (special['kernel_of']() if 'kernel_of' in special else special['kernel']) if special else info.get('kernel', 'decode_trained')
        traffic, traffic_source = live_traffic(workload, kernel_name, args.cache_dir)
        if traffic is None:
            why_not_live, traffic_source = traffic_source, None
    if traffic is None and not strong_main and recorded.get(workload) is not None:
        traffic = recorded[workload]
        traffic_source = 'static: profiles/hbm_traffic.json ({}){}'.format(
            recorded.get('_source', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/perf/prof.sh'),
            '; live passes: ' + why_not_live if why_not_live else '')
    traffic_recorded = recorded.get(workload) if not strong_main else None

    total_words = sum(entry['batch'] for entry in per_rank)
    result = {
        'metric': 'embeddings/sec (and HBM GB/s vs roofline), 300-dim {}-bit batch lookup'.format(bits),
        'value': total_words * args.steps / elapsed,
        'unit': 'embeddings/s',
        'n_gpus': args.gpus,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': elapsed / args.steps * 1e3,
        'higher_is_better': True,
        'scaling': 'strong' if strong_main else 'weak',
        'vs_baseline': None,
        'dtype': 'u32',
        'data': 'synthetic',
        'config': dict({
            'workload': workload,
            'vocabulary': count,
            'dim': dim,
            'storage': 'trained',
            'bits_per_weight': bits,
            'batch_per_gpu': n,
            'batch': ('keys() full dump' if batch is None else 'uniform random rows, 1% misses, seed 11') +
                     (', ONE batch split over the ranks as sharding.shard_range does' if strong_main else '') +
                     (', and three more batches like it (seeds 12-14): four batches round-robin into four buffers, nothing cached between launches'
                      if special is None and rotation is not None else ''),
            'vectors': 'N(0, 0.4^2) seed 1234, written by memb_amd.Builder',
            'parallelism': 'batch shards, model replicated per GPU, no collective',
        }, **(special['config'] if special else {})),
        'roofline': {
            'bound': 'hbm',
            'achieved': achieved_gbps,
            'peak': HBM_PEAK_GBPS,
            'unit': 'GB/s',
            'frac': achieved_gbps / HBM_PEAK_GBPS,
            'traffic': traffic,
            'traffic_source': traffic_source,
            'traffic_over_algorithmic': (traffic / nbytes) if traffic else None,
            'traffic_recorded_in_profiles': traffic_recorded,   # profiles/hbm_traffic.json (the builder's prof.sh passes), for comparison
            'kernel': (special['kernel_of']() if 'kernel_of' in special else special['kernel']) if special else info.get('kernel', 'decode_trained'),
            'kernel_avg_ms': kernel_avg_ms,
            'kernel_timing': kernel_timing,
            'kernel_min_ms': kernel_min_ms,
            'kernel_median_ms': kernel_median_ms,
            'kernel_ms_in_launch_order': [round(starts[i].elapsed_time(stops[i]), 4) for i in range(args.steps)],
            'algorithmic_bytes_per_launch': nbytes,
            'algorithmic_bytes_per_word': nbytes / max(n, 1),
            'frac_of_copy_ceiling': achieved_gbps / HBM_COPY_CEILING_GBPS,
            'box_fill': {
                'what': 'torch fill_ of the same output buffer on this GPU, {} launches enqueued in front of the warmup steps, median of the second half (write-only yardstick, not a bound: boxes differ by ~10 %; also the run-in of the power state, see DESIGN.md section 6)'.format(FILL_LAUNCHES),
                'ms': fill_ms[len(fill_ms) // 2],
                'GBps': 4.0 * n * dim / (fill_ms[len(fill_ms) // 2] * 1e-3) / 1e9,
                'kernel_hbm_bytes_rate_vs_fill': ((traffic or nbytes) / (kernel_avg_ms * 1e-3)) / (4.0 * n * dim / (fill_ms[len(fill_ms) // 2] * 1e-3)),
            },
            'box_ceilings': ceilings,
            'rank': 0,
        },
        'cpu_baseline': baseline,
        'parity_vs_cpu_checker': parity,
        'ranks_seen': len(per_rank),
        'rehearsal': ('cpu: host stand-ins for the device (install_host_stand_ins) -- plumbing only, no number in this line is a measurement' if cpu_rehearsal
                      else 'every rank on cuda:0, gloo rendezvous' if rehearsal else None),
        'launcher': json.loads(os.environ['MEMB_BENCH_LAUNCHER']) if os.environ.get('MEMB_BENCH_LAUNCHER') else
                    {'started_by': 'torch.distributed.run' if 'TORCHELASTIC_RUN_ID' in os.environ else 'python bench.py'},
        'per_rank': per_rank,
        'strong_scaling': strong,
        'configs': configs,
        'host_api': host_api,
        'word_search': word_search,
        'kernel_embeddings_per_s': n / (kernel_avg_ms * 1e-3),
        'geometry': {k: info.get(k) for k in ('waves_per_block', 'tiles_per_wavefront', 'lanes_per_word', 'segment_symbols', 'lds_bytes_per_block', 'root_bits',
                                              'max_code_bits', 'max_stream_bytes', 'device_bytes', 'word_index_bytes', 'row_layout', 'row_bytes')},
        'model_build_s': build_seconds,
        'model_writer': 'host (memb_amd.Builder)' if args.host_writer else 'device (memb_amd.Builder(device={}): memb_hip_encoder_*)'.format(local_rank),
        'reader_open_s': open_seconds,
    }
    print(json.dumps(result))
    sys.stdout.flush()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
