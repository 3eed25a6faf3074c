#!/usr/bin/env python3
"""Benchmark of the memb batch-lookup hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path over one batch: every word of a synthetic
GloVe-840B-shaped model (2,196,017 words x 300, trained 4-bit) is looked up
once, row ids and the fp32 output resident in HBM (BASELINE.json north_star:
">= 50 % of HBM bandwidth on a 2.2M-word x 300-dim 4-bit batch lookup").
With N > 1 (one process per GPU, launched by torch.distributed.run) every rank
holds a replica of the model and looks up its own full-size batch -- shards of
a vocabulary N times as large -- with no collective on the data path (weak
scaling); value = words all ranks decoded / max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
HBM_COPY_CEILING_GBPS = 6290.0  # measured float4 copy on MI355X (same guide)

WORKLOADS = {
    # name: (words, bits, batch) ; batch None = every key (full dump)
    'glove840b-300d-4bit-fullvocab': (2196017, 4, None),     # north-star headline (SURVEY 8d "H")
    'glove840b-300d-4bit-100k': (2196017, 4, 100000),        # BASELINE.json configs[1]
    'fasttext2m-300d-6bit-fullvocab': (1999995, 6, None),    # BASELINE.json configs[2]
    'glove840b-300d-2bit-fullvocab': (2196017, 2, None),     # BASELINE.json configs[3] (per GPU)
    'small-4bit': (50000, 4, None),                          # quick functional run
}


def parse_args():
    parser = argparse.ArgumentParser()
    parser.add_argument('--gpus', type=int, default=1)
    parser.add_argument('--steps', type=int, default=20)
    parser.add_argument('--warmup', type=int, default=5)
    parser.add_argument('--workload', default='glove840b-300d-4bit-fullvocab', choices=sorted(WORKLOADS))
    parser.add_argument('--cache-dir', default=os.environ.get('MEMB_BENCH_CACHE', '/tmp/memb_amd_bench'))
    parser.add_argument('--no-cpu-baseline', action='store_true')
    parser.add_argument('--secondary', action='store_true',
                        help='also time the 100k-row batch of configs[1] (extra launches of the same kernel)')
    return parser.parse_args()


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
    """reader[words] -> numpy for the whole batch and for 100 000 of its words, best of 3; and the CPU
    restatement's Reader.batch_embedding (word search + decode, all host threads) on the 100 000."""
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
    checker = oracle.OracleReader(path, os.cpu_count() or 1)
    cpu_part = best_of(lambda: checker.batch_embedding(sample))
    return {
        'note': 'words in, numpy float32 out (word search, PCIe, host memory, result allocation included); never part of value',
        'batch_words': len(words),
        'batch_seconds': whole,
        'batch_embeddings_per_s': len(words) / whole,
        'sample_words': len(sample),
        'sample_seconds': part,
        'sample_embeddings_per_s': len(sample) / part,
        'cpu_port_sample_seconds': cpu_part,
        'cpu_port_sample_embeddings_per_s': len(sample) / cpu_part,
    }


def main():
    args = parse_args()
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world_size = int(os.environ.get('WORLD_SIZE', '1'))
    if world_size != args.gpus:
        if world_size == 1 and args.gpus > 1:
            raise SystemExit('--gpus {} needs torch.distributed.run with one process per GPU'.format(args.gpus))
        args.gpus = world_size

    import numpy as np
    import torch
    import torch.distributed as dist

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    distributed = world_size > 1
    # Rehearsal of the multi-rank plumbing on a one-GPU box: MEMB_BENCH_REHEARSAL=1 puts every rank on
    # cuda:0 and uses gloo (RCCL refuses two ranks on one device). Never set by the driver.
    rehearsal = distributed and os.environ.get('MEMB_BENCH_REHEARSAL') == '1'
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if rehearsal:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        dist.barrier()
    import memb_amd

    words, bits, batch = WORKLOADS[args.workload]
    from memb_amd import synthetic
    os.environ['MEMB_BENCH_CACHE'] = args.cache_dir
    build_seconds = 0.0
    if rank == 0:
        path, build_seconds = synthetic.cached_model(words, 300, 'trained', bits)   # written once per box
    if distributed:
        dist.barrier()
    path, _ = synthetic.cached_model(words, 300, 'trained', bits)

    open_start = time.time()
    reader = memb_amd.Reader(path, device=local_rank)
    info = reader.info()   # stages the model to HBM
    open_seconds = time.time() - open_start
    dim = reader.dim
    count = len(reader)

    if batch is None:
        rows_host = np.arange(count, dtype=np.uint32)   # batch = keys(): rows in sorted-word order
    else:
        rng = np.random.default_rng(11)
        rows_host = rng.integers(0, count, size=batch).astype(np.uint32)
        rows_host[rng.integers(0, batch, size=batch // 100)] = 0xFFFFFFFF   # 1 % misses
    n = len(rows_host)

    import ctypes
    library = ctypes.CDLL(memb_amd.HIP_LIBRARY_PATH)
    algorithmic = ctypes.c_uint64(0)
    library.memb_hip_algorithmic_bytes(
        ctypes.c_void_p(reader._impl.context_handle()), rows_host.ctypes.data_as(ctypes.c_void_p),
        ctypes.c_size_t(n), ctypes.byref(algorithmic))
    algorithmic_bytes = algorithmic.value

    rows = torch.from_numpy(rows_host.view(np.int32)).cuda()
    out = torch.empty((n, dim), dtype=torch.float32, device='cuda')

    def step():
        reader.rows_embedding_device(rows, out=out)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    # per-launch kernel durations from HIP events on the stream the kernel runs on
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    stops = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
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
    kernel_avg_ms = sum(kernel_ms) / len(kernel_ms)

    if rank != 0:
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return

    # BASELINE.json configs[1] on the same model: a 100 000-row random batch with 1 % misses
    # (rank 0 only, outside the timed region; kernel time from HIP events).
    secondary = None
    if batch is None and args.secondary:
        rng = np.random.default_rng(11)
        small_host = rng.integers(0, count, size=100000).astype(np.uint32)
        small_host[rng.integers(0, 100000, size=1000)] = 0xFFFFFFFF
        small_rows = torch.from_numpy(small_host.view(np.int32)).cuda()
        small_out = torch.empty((100000, dim), dtype=torch.float32, device='cuda')
        for _ in range(5):
            reader.rows_embedding_device(small_rows, out=small_out)
        torch.cuda.synchronize()
        events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
        for begin, end in events:
            begin.record()
            reader.rows_embedding_device(small_rows, out=small_out)
            end.record()
        torch.cuda.synchronize()
        small_ms = sorted(begin.elapsed_time(end) for begin, end in events)
        small_bytes = ctypes.c_uint64(0)
        library.memb_hip_algorithmic_bytes(
            ctypes.c_void_p(reader._impl.context_handle()), small_host.ctypes.data_as(ctypes.c_void_p),
            ctypes.c_size_t(len(small_host)), ctypes.byref(small_bytes))
        median = small_ms[len(small_ms) // 2]
        secondary = {
            'workload': 'glove840b-300d-4bit-100k (BASELINE.json configs[1])',
            'kernel_median_ms': median,
            'embeddings_per_s': 100000 / (median * 1e-3),
            'algorithmic_GBps': small_bytes.value / (median * 1e-3) / 1e9,
        }

    # parity spot check of the timed output against the CPU checker, and the CPU baseline
    baseline = None
    parity = 'skipped'
    # (the CPU legs run at N = 1 only: with more ranks the others would wait at the final barrier)
    cpu_legs = not args.no_cpu_baseline and world_size == 1
    if cpu_legs:
        baseline, expected = cpu_baseline(path, rows_host, dim)
        got = out.cpu().numpy()
        parity = 'bit-exact' if np.array_equal(got.view(np.uint32), expected.view(np.uint32)) else 'MISMATCH'
    elif not args.no_cpu_baseline:
        # N > 1: no timed CPU leg, but rank 0's output is still checked on a sample of the batch
        import oracle
        sample = min(n, 20000)
        expected = oracle.OracleReader(path, os.cpu_count() or 1).rows_embedding(rows_host[:sample])
        got = out[:sample].cpu().numpy()
        parity = ('bit-exact (first {} rows)'.format(sample)
                  if np.array_equal(got.view(np.uint32), expected.view(np.uint32)) else 'MISMATCH')

    # Not part of `value`: what a caller of the reference's API sees (words in, numpy out; word search,
    # PCIe and host memory included), next to the restated CPU Reader on the same words and host cores.
    host_api = None
    if cpu_legs:
        host_api = host_api_timings(reader, path, rows_host)

    achieved_gbps = algorithmic_bytes / (kernel_avg_ms * 1e-3) / 1e9
    traffic = None
    traffic_file = os.path.join(REPO, 'profiles', 'hbm_traffic.json')
    if os.path.exists(traffic_file):
        with open(traffic_file) as f:
            traffic = json.load(f).get(args.workload)

    result = {
        'metric': 'embeddings/sec (and HBM GB/s vs roofline), 300-dim {}-bit batch lookup'.format(bits),
        'value': args.gpus * n * args.steps / elapsed,
        'unit': 'embeddings/s',
        'n_gpus': args.gpus,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': elapsed / args.steps * 1e3,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'u32',
        'data': 'synthetic',
        'config': {
            'workload': args.workload,
            'vocabulary': count,
            'dim': dim,
            'storage': 'trained',
            'bits_per_weight': bits,
            'batch_per_gpu': n,
            'batch': 'keys() full dump' if batch is None else 'uniform random rows, 1% misses, seed 11',
            'vectors': 'N(0, 0.4^2) seed 1234, written by memb_amd.Builder',
            'parallelism': 'batch shards, model replicated per GPU, no collective',
        },
        'roofline': {
            'bound': 'hbm',
            'achieved': achieved_gbps,
            'peak': HBM_PEAK_GBPS,
            'unit': 'GB/s',
            'frac': achieved_gbps / HBM_PEAK_GBPS,
            'traffic': traffic,
            'kernel': 'decode_trained_persistent',
            'kernel_avg_ms': kernel_avg_ms,
            'kernel_min_ms': kernel_ms[0],
            'algorithmic_bytes_per_launch': algorithmic_bytes,
            'algorithmic_bytes_per_word': algorithmic_bytes / n,
            'frac_of_copy_ceiling': achieved_gbps / HBM_COPY_CEILING_GBPS,
        },
        'cpu_baseline': baseline,
        'parity_vs_cpu_checker': parity,
        'host_api': host_api,
        'kernel_embeddings_per_s': n / (kernel_avg_ms * 1e-3),
        'secondary': secondary,
        'geometry': {k: info[k] for k in ('waves_per_block', 'lanes_per_word', 'segment_symbols', 'lds_bytes_per_block', 'root_bits',
                                          'max_code_bits', 'max_stream_bytes', 'device_bytes')},
        'model_build_s': build_seconds,
        'reader_open_s': open_seconds,
    }
    print(json.dumps(result))
    sys.stdout.flush()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
