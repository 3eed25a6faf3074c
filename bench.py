#!/usr/bin/env python3
"""Benchmark of the memb batch-lookup hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path over one batch: every word of a synthetic GloVe-840B-shaped model
(2,196,017 words x 300, trained 4-bit) is looked up once, row ids and the fp32 output resident in HBM
(BASELINE.json north_star: ">= 50 % of HBM bandwidth on a 2.2M-word x 300-dim 4-bit batch lookup").

N > 1: one process per GPU, every rank a replica of the model, no collective on the data path. Started by
torch.distributed.run the script is a rank; started plainly as `python bench.py --gpus N` it first spawns the
N ranks as child processes (before it makes any GPU call itself) and relays rank 0's line. The main line is
weak scaling (one full-size batch per rank); `strong_scaling` = BASELINE.json configs[3], ONE 2-bit dump
split N ways as memb_amd.sharding.shard_range does (the reference's own split, src/reader.cpp:65-79).

Rank 0 writes ONE line to stdout: a compact JSON object, the record. It stays below 8 KB
(tests/test_bench_contract.py; the driver's record keeps the last 8 081 characters of stdout): the required keys,
`roofline`, `cpu_baseline` and one terse entry per BASELINE.json configuration. Everything measured, verbose, goes
to stderr (`detail: {...}`) and to gpurun_out/bench_detail.json. Ceilings, word search and host-API legs live in
tools/perf/bench_extras.py (`--extras` runs them into the detail).
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for directory in (os.path.join(REPO, 'tools', 'perf'), REPO):
    if directory not in sys.path:
        sys.path.insert(0, directory)
from bench_support import Timer, hip_runtime_mapped, kfd_gpu_count, live_traffic, spawn_ranks, prebuild_models, recorded_traffic, sources_sha16   # noqa: E402 (no torch, no HIP in there)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
GLOVE_WORDS = 2196017
FASTTEXT_WORDS = 1999995
MISSING = 0xFFFFFFFF
FILL_LAUNCHES = 40
LINE_LIMIT = 8000               # the driver's record keeps the last 8 081 characters of stdout

WORKLOADS = {
    # name: (words, bits, batch) ; batch None = every key (full dump)
    'glove840b-300d-4bit-fullvocab': (GLOVE_WORDS, 4, None),     # north-star headline (SURVEY 8d "H")
    'glove840b-300d-4bit-100k': (GLOVE_WORDS, 4, 100000),        # BASELINE.json configs[1]
    'fasttext2m-300d-6bit-fullvocab': (FASTTEXT_WORDS, 6, None),  # BASELINE.json configs[2]
    'glove840b-300d-2bit-fullvocab': (GLOVE_WORDS, 2, None),     # BASELINE.json configs[3]
    'small-4bit': (50000, 4, None),                              # quick functional run
    # not single trained models: the step comes from tools/perf/bench_extras.special_workload (profiling runs)
    'union-concat-500k': (GLOVE_WORDS, 4, 500000),               # BASELINE.json configs[4]
    'uniform-8bit-500k': (500000, 8, None),
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
    parser.add_argument('--no-configs', action='store_true', help='only the timed kernel (profiling runs)')
    parser.add_argument('--no-live-traffic', action='store_true', help='no rocprofv3 --pmc child passes for roofline.traffic')
    parser.add_argument('--extras', action='store_true', help='also tools/perf/bench_extras.py legs (ceilings, word search, host API) into the detail line')
    parser.add_argument('--small', action='store_true', help='shrink every model to 50 000 words (plumbing rehearsal)')
    parser.add_argument('--host-writer', action='store_true', help='write the synthetic models with the host writer (same bytes)')
    parser.add_argument('--dry-launch', action='store_true', help='--gpus N: print the launch command and what the parent saw; start nothing')
    return parser.parse_args()


# --------------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------------

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
    same = np.array_equal(np.ascontiguousarray(got_rows(picks)).view(np.uint32), expected.view(np.uint32))
    return 'bit-exact ({} sampled rows)'.format(len(picks)) if same else 'MISMATCH'


def batch_rows(count, batch, np, seed=11):
    if batch is None:
        return np.arange(count, dtype=np.uint32)   # batch = keys(): rows in sorted-word order
    rng = np.random.default_rng(seed)
    rows = rng.integers(0, count, size=batch).astype(np.uint32)
    rows[rng.integers(0, batch, size=batch // 100)] = MISSING   # 1 % misses
    return rows


def cpu_baseline(path, rows_host, dim):
    """CPU decode of the same batch on this box's host cores, pre-resolved rows, decode only: the reference's own
    HuffmanTableDecoder + centroid gather (oracle/_ref, kind "reference") split over threads as
    Reader::batchEmbeddingToBuffer splits a batch, or oracle/memb_oracle.c (kind "port"). Its output is also the
    parity check of the timed GPU result."""
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
    out = np.empty((len(rows_host), dim), dtype=np.float32)
    best, passes, deadline = float('inf'), 0, time.time() + 6.0
    while passes < 3 or time.time() < deadline:
        start = time.time()
        decode(rows_host, out, cores)
        best = min(best, time.time() - start)
        passes += 1
    single = rows_host[:min(len(rows_host), 100000)]
    start = time.time()
    decode(single, out[:len(single)], 1)
    single_rate = len(single) / (time.time() - start)
    if kind == 'reference':   # the restatement must agree with the reference on this batch too
        port = reader.rows_embedding(single, num_threads=cores)
        if not np.array_equal(port.view(np.uint32), out[:len(single)].view(np.uint32)):
            raise SystemExit('oracle/memb_oracle.c and oracle/_ref disagree')
    return {'value': len(rows_host) / best, 'unit': 'embeddings/s', 'cores': cores, 'kind': kind,
            'sample': '{} pre-resolved rows of the same batch, decode only, {} threads, best of {} passes; 1 thread: {:.0f}/s'.format(
                len(rows_host), cores, passes, single_rate)}, out


def open_reader(memb_amd, path, device, batch_words=0):
    """Open + stage on this rank's GPU: the model and the word -> row index (what a rank that serves words pays)."""
    start = time.time()
    reader = memb_amd.Reader(path, device=device)
    reader.info(batch_words)   # stages the model to HBM
    if os.environ.get('MEMB_BENCH_REHEARSAL') != 'cpu':
        reader.stage_words()
    return reader, reader.info(batch_words), time.time() - start


# --------------------------------------------------------------------------------------------
# the configurations of BASELINE.json: kernel time, algorithmic bytes, parity sample
# --------------------------------------------------------------------------------------------

def measure_config(name, reader, path, rows_host, timer, library, torch, np, launches=15, rotate=0):
    """One configuration. rotate = K > 0: K different batches of this size round-robin into K output buffers, so that
    nothing of a launch is still in the 256 MB Infinity Cache at its next turn (`frac` is then that HBM-regime figure
    and `repeated_buffer_frac` the cache-assisted one of one batch re-decoded into one buffer)."""
    rows = torch.from_numpy(rows_host.view(np.int32)).cuda()
    out = torch.empty((len(rows_host), reader.dim), dtype=torch.float32, device='cuda')
    ms, how = timer.median_ms(lambda: reader.rows_embedding_device(rows, out=out), launches)
    nbytes = algorithmic_bytes(library, reader, rows_host)
    entry = {'workload': name, 'kernel': reader.info(len(rows_host)).get('kernel', ''), 'batch': len(rows_host),
             'kernel_ms': ms, 'frac': nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 'algorithmic_bytes': nbytes, 'timing': how,
             'parity': sampled_parity(path, rows_host, lambda picks: out[torch.from_numpy(picks).cuda()].cpu().numpy())}
    if rotate:
        sets = [(rows_host, rows, out)]
        for k in range(1, rotate):
            other = batch_rows(len(reader), len(rows_host), np, seed=110 + k)
            sets.append((other, torch.from_numpy(other.view(np.int32)).cuda(), torch.empty_like(out)))
        turn = [0]

        def call():
            _, ids, target = sets[turn[0] % rotate]
            turn[0] += 1
            reader.rows_embedding_device(ids, out=target)

        averages = timer.bursts(call, 15 * rotate)
        hbm_ms = averages[len(averages) // 2]
        hbm_bytes = sum(algorithmic_bytes(library, reader, host) for host, _, _ in sets) / rotate
        entry.update({'repeated_buffer_frac': entry['frac'], 'repeated_buffer_ms': ms, 'kernel_ms': hbm_ms,
                      'frac': hbm_bytes / (hbm_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 'algorithmic_bytes': int(hbm_bytes),
                      'timing': '{} batches round-robin into {} buffers, nothing cached between launches; bursts, one HIP event pair each'.format(rotate, rotate)})
    return entry


def measure_union(reader_a, path_a, reader_b, path_b, timer, library, torch, np, batch=500000, launches=15):
    """BASELINE.json configs[4]: ReadersUnion 'concatenate' of two 4-bit models, 500 000 words, (n, 600) output;
    a quarter of the words is missing from each model."""
    import oracle
    from memb_amd import _memb
    rng = np.random.default_rng(17)
    rows_a = rng.integers(0, len(reader_a), size=batch).astype(np.uint32)
    rows_a[rng.random(batch) < 0.25] = MISSING
    rows_b = rng.integers(0, len(reader_b), size=batch).astype(np.uint32)
    rows_b[rng.random(batch) < 0.25] = MISSING
    ids = [torch.from_numpy(rows_a.view(np.int32)).cuda(), torch.from_numpy(rows_b.view(np.int32)).cuda()]
    merged = torch.empty((batch, reader_a.dim + reader_b.dim), dtype=torch.float32, device='cuda')
    stream = torch.cuda.current_stream().cuda_stream

    def fused():
        return _memb.union_rows_to_device([reader_a._impl, reader_b._impl], [ids[0].data_ptr(), ids[1].data_ptr()], [0, reader_a.dim],
                                          batch, merged.data_ptr(), merged.stride(0), stream, False)

    def per_reader():   # models of different key formats cannot share the kernel: one launch per column block
        reader_a.rows_embedding_device(ids[0], out=merged, col_off=0)
        reader_b.rows_embedding_device(ids[1], out=merged, col_off=reader_a.dim)

    one_launch = bool(fused())
    ms, how = timer.median_ms(fused if one_launch else per_reader, launches)
    nbytes = algorithmic_bytes(library, reader_a, rows_a) + algorithmic_bytes(library, reader_b, rows_b) - 4 * batch
    picks = np.sort(rng.choice(batch, size=20000, replace=False))
    cores = os.cpu_count() or 1
    expected = np.concatenate([oracle.OracleReader(path_a, cores).rows_embedding(np.ascontiguousarray(rows_a[picks])),
                               oracle.OracleReader(path_b, cores).rows_embedding(np.ascontiguousarray(rows_b[picks]))], axis=-1)
    got = merged[torch.from_numpy(picks).cuda()].cpu().numpy()
    return {'workload': 'union-concat-glove4bit+fasttext4bit-500k (BASELINE.json configs[4])',
            'kernel': (reader_a.info().get('union_kernel') or 'fused union kernel') if one_launch else 'decode_trained x 2',
            'batch': batch, 'kernel_ms': ms, 'frac': nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 'algorithmic_bytes': nbytes, 'timing': how,
            'parity': 'bit-exact (20000 sampled rows)' if np.array_equal(got.view(np.uint32), expected.view(np.uint32)) else 'MISMATCH'}


def all_configs(args, memb_amd, synthetic, headline, timer, library, torch, np, sizes):
    """Every BASELINE.json configuration (and the two full-size batches in shuffled order), on cuda:0, outside the timed region."""
    glove, fasttext = sizes
    reader4, path4 = headline
    results = []
    build_seconds = 0.0

    def model(*spec, **more):
        nonlocal build_seconds
        path, spent = synthetic.cached_model(*spec, **more)
        build_seconds += spent
        return memb_amd.Reader(path, device=0), path

    reader, path = model(1000, 300, 'uniform', 8)   # configs[0]: the reference's CPU-runnable plumbing case, here through the HIP path
    rows = np.concatenate([np.arange(1000, dtype=np.uint32), np.full(100, MISSING, dtype=np.uint32)])
    results.append(measure_config('uniform-8bit-1k (BASELINE.json configs[0])', reader, path, rows, timer, library, torch, np))
    results.append(measure_config('glove840b-300d-4bit-100k (BASELINE.json configs[1])', reader4, path4,
                                  batch_rows(len(reader4), 100000, np), timer, library, torch, np, launches=30, rotate=4))
    # the headline's batch in RANDOM order (token streams are: reference src/reader.cpp:49-57 takes words in any order)
    results.append(measure_config('glove840b-300d-4bit-fullvocab-shuffled', reader4, path4,
                                  np.random.default_rng(77).permutation(len(reader4)).astype(np.uint32), timer, library, torch, np))
    reader, path = model(fasttext, 300, 'trained', 6)
    results.append(measure_config('fasttext2m-300d-6bit-fullvocab (BASELINE.json configs[2])', reader, path,
                                  np.arange(len(reader), dtype=np.uint32), timer, library, torch, np))
    results.append(measure_config('fasttext2m-300d-6bit-fullvocab-shuffled', reader, path,
                                  np.random.default_rng(78).permutation(len(reader)).astype(np.uint32), timer, library, torch, np))
    reader, path = model(glove, 300, 'trained', 2)
    results.append(measure_config('glove840b-300d-2bit-fullvocab (BASELINE.json configs[3], one GPU: the whole dump)', reader, path,
                                  np.arange(len(reader), dtype=np.uint32), timer, library, torch, np))
    reader, path = model(fasttext, 300, 'trained', 4, seed=4321)
    results.append(measure_union(reader4, path4, reader, path, timer, library, torch, np, batch=min(500000, len(reader4))))
    reader, path = model(min(500000, glove), 300, 'uniform', 8)
    results.append(measure_config('uniform-8bit-500k', reader, path, np.arange(len(reader), dtype=np.uint32), timer, library, torch, np))
    del reader
    recorded, source = recorded_traffic()
    for entry in results:
        key = entry['workload'].split(' ')[0].replace('union-concat-glove4bit+fasttext4bit-500k', 'union-concat-500k')
        traffic = None if args.small else recorded.get(key)
        entry['traffic'] = traffic
        entry['traffic_over_algorithmic'] = traffic / entry['algorithmic_bytes'] if traffic else None
        entry['traffic_source'] = source if traffic or source == 'stale' else None
    return results, build_seconds


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

    def with_d2h():   # each rank also copies its slice into its own pinned host buffer: the disjoint slices are the host-side gather
        reader.rows_embedding_device(rows, out=out)
        host.copy_(out, non_blocking=True)

    elapsed_kernel = timed(kernel_only)
    kernel_ms = Timer(torch).launches(kernel_only, steps)
    elapsed_d2h = timed(with_d2h)
    # the product's own host-side gather: every rank decodes its slice straight into its rows of a host matrix
    host_rows = np.zeros((len(mine), reader.dim), dtype=np.float32)
    elapsed_host = timed(lambda: reader.rows_embedding_into(mine, host_rows))
    host_parity = sampled_parity(path, mine, lambda picks: host_rows[picks], sample=5000)
    parity = sampled_parity(path, mine, lambda picks: host[torch.from_numpy(picks)].numpy(), sample=5000)
    summary = {'rank': rank, 'device': local_rank, 'rows': [int(start), int(stop)],
               'kernel_avg_ms': round(sum(kernel_ms) / len(kernel_ms), 5), 'parity': parity}
    if distributed:
        gathered = [None] * world_size
        dist.all_gather_object(gathered, summary)
    else:
        gathered = [summary]
    del reader, rows, out, host
    return {'workload': name + ' (BASELINE.json configs[3]): ONE dump split over the ranks, no collective', 'scaling': 'strong',
            'n_gpus': world_size, 'ranks_seen': len(gathered), 'steps': steps,
            'kernel_only': {'value': count * steps / elapsed_kernel, 'unit': 'embeddings/s', 'ms_per_step': elapsed_kernel / steps * 1e3},
            'with_d2h': {'value': count * steps / elapsed_d2h, 'unit': 'embeddings/s', 'ms_per_step': elapsed_d2h / steps * 1e3},
            'host_gather': {'value': count * steps / elapsed_host, 'unit': 'embeddings/s', 'ms_per_step': elapsed_host / steps * 1e3,
                            'parity_rank0': host_parity},
            'per_rank': gathered}, build_seconds


# --------------------------------------------------------------------------------------------
# the record
# --------------------------------------------------------------------------------------------

CONFIG_KEYS = ('workload', 'kernel', 'batch', 'kernel_ms', 'frac', 'repeated_buffer_frac', 'traffic_over_algorithmic', 'traffic_source', 'parity')


def rounded(value, digits=6, text=160):
    """Floats to `digits` significant digits, strings to `text` characters (the record is a record, not documentation)."""
    if isinstance(value, float):
        return float('{:.{}g}'.format(value, digits))
    if isinstance(value, str):
        return value if len(value) <= text else value[:text - 3] + '...'
    if isinstance(value, dict):
        return {key: rounded(item, digits, text) for key, item in value.items()}
    if isinstance(value, (list, tuple)):
        return [rounded(item, digits, text) for item in value]
    return value


def compact_line(result):
    """The record: `result` without its verbose parts, as one JSON line of fewer than LINE_LIMIT characters. Never raises for
    length: a record that would not fit loses precision, then optional keys, in that order -- a run that measured everything
    must not end without its line."""
    line = {key: value for key, value in result.items() if key not in ('extras', 'geometry') and value is not None}
    if line.get('configs'):
        line['configs'] = [{key: entry[key] for key in CONFIG_KEYS if entry.get(key) is not None} for entry in line['configs']]
    line['roofline'] = {key: value for key, value in result['roofline'].items() if key not in ('kernel_ms_in_launch_order', 'box_fill')}
    for key in ('vs_baseline', 'cpu_baseline'):   # (contract keys stay even when null)
        line.setdefault(key, None)
    text = json.dumps(rounded(line), separators=(',', ':'))
    for digits, cap, dropped in ((5, 80, ()), (4, 48, ('launcher', 'rehearsal', 'sources_sha16', 'model_build_s', 'reader_open_s')),
                                 (4, 32, ('per_rank', 'kernel_embeddings_per_s')), (4, 24, ('strong_scaling', 'configs'))):
        if len(text) < LINE_LIMIT:
            break
        for key in dropped:
            line.pop(key, None)
        text = json.dumps(rounded(line, digits, cap), separators=(',', ':'))
    return text


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
    # Rehearsals of the multi-rank plumbing (never set by the driver): MEMB_BENCH_REHEARSAL=1 puts every rank on cuda:0 with
    # a gloo rendezvous (RCCL refuses two ranks on one device); =cpu replaces the device by host stand-ins (the CPU test suite).
    cpu_rehearsal = os.environ.get('MEMB_BENCH_REHEARSAL') == 'cpu'
    rehearsal = distributed and os.environ.get('MEMB_BENCH_REHEARSAL') in ('1', 'cpu')
    if cpu_rehearsal:
        import memb_amd
        import bench_rehearsal
        bench_rehearsal.install_host_stand_ins(torch, memb_amd)
        args.host_writer = True
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
    # the synthetic models are written through the device writer (byte-identical files: tests/test_gpu_writer.py) unless asked otherwise
    if args.host_writer:
        os.environ.pop('MEMB_SYNTH_DEVICE', None)
    else:
        os.environ['MEMB_SYNTH_DEVICE'] = str(local_rank)
    build_seconds = 0.0
    if rank == 0:
        needed = [(words, 300, 'trained', bits, 1234)] if workload not in SPECIAL_WORKLOADS else []
        if workload == 'union-concat-500k':
            needed += [(glove, 300, 'trained', 4, 1234), (fasttext, 300, 'trained', 4, 4321)]
        if workload == 'uniform-8bit-500k':
            needed += [(min(500000, glove), 300, 'uniform', 8, 1234)]
        if not args.no_configs and workload not in SPECIAL_WORKLOADS:
            if world_size == 1:
                needed += [(glove, 300, 'trained', 4, 1234), (fasttext, 300, 'trained', 6, 1234), (glove, 300, 'trained', 2, 1234),
                           (fasttext, 300, 'trained', 4, 4321), (min(500000, glove), 300, 'uniform', 8, 1234), (1000, 300, 'uniform', 8, 1234)]
            else:
                needed += [(glove, 300, 'trained', 2, 1234)]
        build_seconds = prebuild_models(synthetic, list(dict.fromkeys(needed)))
    library = ctypes.CDLL(memb_amd.HIP_LIBRARY_PATH)
    timer = Timer(torch)
    special = None
    rotation = None
    if workload in SPECIAL_WORKLOADS:
        if distributed or strong_main:
            raise SystemExit('--workload {} is a one-GPU profiling run'.format(workload))
        args.no_configs = args.no_cpu_baseline = True
        import bench_extras
        special = bench_extras.special_workload(workload, sys.modules[__name__], memb_amd, synthetic, library, torch, np, glove, fasttext)
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
            # A batch whose output fits the 256 MB Infinity Cache: FOUR different batches round-robin into four buffers, so
            # that the timed (and profiled) launches read from and write to HBM; the first is the one whose output is checked.
            rotation = [(rows, out)]
            rotating_bytes = [nbytes]
            for seed in (12, 13, 14):
                other = batch_rows(count, n, np, seed=seed)
                rotating_bytes.append(algorithmic_bytes(library, reader, other))
                rotation.append((torch.from_numpy(other.view(np.int32)).cuda(), torch.empty((n, dim), dtype=torch.float32, device='cuda')))
            nbytes = sum(rotating_bytes) // len(rotating_bytes)
            turn = [0]

            def step():
                ids, target = rotation[turn[0] % len(rotation)]
                turn[0] += 1
                reader.rows_embedding_device(ids, out=target)
        else:
            def step():
                reader.rows_embedding_device(rows, out=out)

    # Events first, so that nothing but launches lies between the phases below.
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    stops = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    fills = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(FILL_LAUNCHES)]
    torch.cuda.synchronize()
    # (1) The box's yardstick and the run-in of its power state: torch's fill_ of the very same output buffer, ~20 ms, directly
    # in front of the warmup (after an idle gap this part runs the first ~25 launches of a burst ~10 % slower: DESIGN.md section 6).
    for begin, end in fills:
        begin.record()
        out.fill_(0.0)
        end.record()
    # (2) W untimed warmup steps
    for _ in range(args.warmup):
        step()
    # (3) the timed region: exactly `steps` steps between barrier + synchronize on both sides; per-launch kernel durations
    # from HIP events on the stream the kernel runs on
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
    in_order = [starts[i].elapsed_time(stops[i]) for i in range(args.steps)]
    kernel_ms = sorted(in_order)
    if special is None:
        info = reader.info(n)   # (kernel and launch geometry are chosen by batch size: a static rule, memb_hip.hip planTrained)
    kernel_avg_ms = sum(kernel_ms) / len(kernel_ms)
    kernel_min_ms, kernel_median_ms = kernel_ms[0], kernel_ms[len(kernel_ms) // 2]
    kernel_timing = 'HIP event pair around every launch of the timed region: average'
    if kernel_avg_ms < 0.2:
        averages = timer.bursts(step, args.steps)
        kernel_avg_ms, kernel_min_ms, kernel_median_ms = sum(averages) / len(averages), averages[0], averages[len(averages) // 2]
        kernel_timing = '5 bursts of {} back-to-back launches after the timed region, one HIP event pair per burst: average per launch'.format(args.steps)
    fill_ms = sorted(begin.elapsed_time(end) for begin, end in fills[FILL_LAUNCHES // 2:])   # the settled half
    rank_summary = {'rank': rank, 'device': local_rank, 'batch': n, 'kernel_avg_ms': round(kernel_avg_ms, 5), 'reader_open_s': round(open_seconds, 3),
                    'device_bytes': info.get('device_bytes'), 'word_index_bytes': info.get('word_index_bytes')}
    if distributed:
        per_rank = [None] * world_size
        dist.all_gather_object(per_rank, rank_summary)
    else:
        per_rank = [rank_summary]

    # BASELINE.json configs[3] split over the ranks (every rank takes part; outside the timed region)
    strong = None
    if not strong_main and not args.no_configs and distributed:
        strong, spent = strong_scaling(args, memb_amd, synthetic, rank, world_size, local_rank, dist, torch, np, library, glove, 2, STRONG_WORKLOAD)
        build_seconds += spent
    if rank != 0:
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return

    # parity of the timed output against the CPU checker, and the CPU baseline (N = 1 only: other ranks would wait at the barrier)
    baseline = None
    parity = 'skipped'
    cpu_legs = not args.no_cpu_baseline and world_size == 1
    if special is None and rotation is not None:
        reader.rows_embedding_device(rows, out=out)   # (the configured batch is the one that is checked)
        torch.cuda.synchronize()
    if cpu_legs:
        baseline, expected = cpu_baseline(path, rows_host, dim)
        parity = 'bit-exact' if np.array_equal(out.cpu().numpy().view(np.uint32), expected.view(np.uint32)) else 'MISMATCH'
        del expected
    elif special is not None:
        parity = special['parity']()
    elif not args.no_cpu_baseline:
        parity = sampled_parity(path, rows_host, lambda picks: out[torch.from_numpy(picks).cuda()].cpu().numpy())

    extras = None
    if args.extras and special is None and world_size == 1 and not cpu_rehearsal:
        import bench_extras
        extras = bench_extras.run(sys.modules[__name__], reader, path, rows_host, out, timer, library, torch, np, kernel_avg_ms)

    configs = None
    if world_size == 1 and not args.no_configs:
        del out, rows
        configs, spent = all_configs(args, memb_amd, synthetic, (reader, path), timer, library, torch, np, (glove, fasttext))
        build_seconds += spent

    achieved_gbps = nbytes / (kernel_avg_ms * 1e-3) / 1e9
    kernel_name = (special['kernel_of']() if 'kernel_of' in special else special['kernel']) if special else info.get('kernel', 'decode_trained')
    traffic, traffic_source = None, None
    if world_size == 1 and not strong_main and not args.no_live_traffic and not args.small and not cpu_rehearsal:
        traffic, traffic_source = live_traffic(workload, kernel_name, args.cache_dir)   # the counters of THIS run
    if traffic is None and not strong_main and not args.small:
        recorded, source = recorded_traffic()
        why_not_live = traffic_source
        traffic, traffic_source = recorded.get(workload), source if recorded.get(workload) or source == 'stale' else None
        if why_not_live and traffic_source:
            traffic_source += ' (live passes: {})'.format(why_not_live)

    result = {
        'metric': 'embeddings/sec (and HBM GB/s vs roofline), 300-dim {}-bit batch lookup'.format(bits),
        'value': sum(entry['batch'] for entry in per_rank) * args.steps / elapsed,
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
            'workload': workload, 'vocabulary': count, 'dim': dim, 'storage': 'trained', 'bits_per_weight': bits, 'batch_per_gpu': n,
            'batch': ('keys() full dump' if batch is None else 'uniform random rows, 1% misses, seed 11') +
                     (', ONE batch split over the ranks (sharding.shard_range)' if strong_main else '') +
                     (', four such batches round-robin into four buffers (nothing cached between launches)' if rotation is not None else ''),
            'vectors': 'N(0, 0.4^2) seed 1234, written by memb_amd.Builder',
            'parallelism': 'batch shards, model replicated per GPU, no collective',
        }, **(special['config'] if special else {})),
        'roofline': {
            'bound': 'hbm', 'achieved': achieved_gbps, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': achieved_gbps / HBM_PEAK_GBPS,
            'traffic': traffic, 'traffic_source': traffic_source, 'traffic_over_algorithmic': (traffic / nbytes) if traffic else None,
            'kernel': kernel_name, 'kernel_avg_ms': kernel_avg_ms, 'kernel_min_ms': kernel_min_ms, 'kernel_median_ms': kernel_median_ms,
            'kernel_timing': kernel_timing, 'algorithmic_bytes_per_launch': nbytes, 'algorithmic_bytes_per_word': nbytes / max(n, 1),
            'kernel_ms_in_launch_order': [round(ms, 4) for ms in in_order],
            'box_fill': {'what': 'torch fill_ of the same output buffer in front of the warmup, median of the settled half',
                         'ms': fill_ms[len(fill_ms) // 2], 'GBps': 4.0 * n * dim / (fill_ms[len(fill_ms) // 2] * 1e-3) / 1e9},
        },
        'cpu_baseline': baseline,
        'parity_vs_cpu_checker': parity,
        'ranks_seen': len(per_rank),
        'rehearsal': ('cpu: host stand-ins for the device -- plumbing only, no number in this line is a measurement' if cpu_rehearsal
                      else 'every rank on cuda:0, gloo rendezvous' if rehearsal else None),
        'launcher': json.loads(os.environ['MEMB_BENCH_LAUNCHER']) if os.environ.get('MEMB_BENCH_LAUNCHER') else
                    {'started_by': 'torch.distributed.run' if 'TORCHELASTIC_RUN_ID' in os.environ else 'python bench.py'},
        'per_rank': per_rank,
        'strong_scaling': strong,
        'configs': configs,
        'kernel_embeddings_per_s': n / (kernel_avg_ms * 1e-3),
        'geometry': {k: info.get(k) for k in ('waves_per_block', 'tiles_per_wavefront', 'lanes_per_word', 'lds_bytes_per_block', 'row_bytes')},
        'sources_sha16': sources_sha16(),
        'model_build_s': round(build_seconds, 2),
        'reader_open_s': round(open_seconds, 3),
        'extras': extras,
    }
    detail = json.dumps(result)
    detail_path = os.environ.get('MEMB_BENCH_DETAIL', os.path.join(REPO, 'gpurun_out', 'bench_detail.json'))
    try:
        os.makedirs(os.path.dirname(detail_path), exist_ok=True)
        with open(detail_path, 'w') as f:
            f.write(detail + '\n')
    except OSError:
        pass
    print('detail: ' + detail, file=sys.stderr)
    sys.stderr.flush()
    print(compact_line(result))   # the record: the ONLY line this script writes to stdout
    sys.stdout.flush()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
