"""The legs of the benchmark that are NOT the record (bench.py --extras puts them into the detail; also runnable alone:
`python tools/perf/bench_extras.py`): the box's own ceilings for the decoder's memory pattern (tools/perf/ceilings.hip),
the word search host vs device, the host API (words in, numpy out), small batches, several batches in one launch, and
the two profiling workloads that are not one trained model (`bench.py --workload union-concat-500k | uniform-8bit-500k`).
`bench` below is the bench.py module (its helpers and constants)."""
import ctypes
import os
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MISSING = 0xFFFFFFFF
HBM_PEAK_GBPS = 8000.0
OFF_PATH_SEED = 99   # N(0, 0.4^2) vectors whose 4-bit code has a 9-bit word: byte keys instead of the nibble-key fast path


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
        # the same words packed already (a tokenizer's output: UTF-8 bytes + n + 1 offsets): no str object is walked
        encoded = [word.encode('utf-8') for word in words]
        blob = b''.join(encoded)
        starts = np.zeros(len(words) + 1, dtype=np.uint32)
        np.cumsum([len(e) for e in encoded], out=starts[1:])
        del encoded
        packed_rows = torch.empty(len(words), dtype=torch.int32, device='cuda')

        def packed():
            reader.resolve_packed_device(blob, starts, out=packed_rows)
            torch.cuda.synchronize()

        packed()
        agree_packed = bool(np.array_equal(packed_rows.cpu().numpy().view(np.uint32), expected))
        packed_s = best(packed)
        packed_fill_s = best(lambda: _memb._packed_fill_seconds(scratch, blob, starts))
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
            'device_ms_from_packed_words': packed_s * 1e3, 'packed_words_to_pinned_memory_alone_ms': packed_fill_s * 1e3, 'packed_parity': 'equal to the host search' if agree_packed else 'MISMATCH',
            'device_breakdown_ms': {'strings -> pinned memory alone (no lookup)': fill_s * 1e3,
                                    'resolve_words over the whole batch alone (reads the words over PCIe)': min(kernel_ms)},
            'words_per_s_device': len(words) / device_s,
            'parity': ('device == host search on every word; host == CPU checker (lower_bound + strcmp) on {} sampled words'.format(len(sample))
                       if agree and agree_checker else 'MISMATCH'),
        })
    return result


def special_workload(name, bench, memb_amd, synthetic, library, torch, np, glove, fasttext):
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
            'nbytes': bench.algorithmic_bytes(library, reader, rows_host), 'kernel': info['kernel'], 'info': info, 'keep': (reader, rows),
            'parity': lambda: bench.sampled_parity(path, rows_host, lambda picks: out[torch.from_numpy(picks).cuda()].cpu().numpy()),
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
        'nbytes': bench.algorithmic_bytes(library, reader_a, rows_a) + bench.algorithmic_bytes(library, reader_b, rows_b) - 4 * batch,
        'kernel': 'decode_union_split', 'kernel_of': lambda: reader_a.info().get('union_kernel') or 'decode_union_split',
        'info': info, 'keep': (reader_a, reader_b, ids), 'parity': parity,
        'config': {'vocabulary': [len(reader_a), len(reader_b)], 'dim': reader_a.dim + reader_b.dim, 'storage': 'trained + trained',
                   'bits_per_weight': 4, 'batch': '500 000 random words, 25 % missing per model, ReadersUnion concatenate in one launch'},
        'build_seconds': build_seconds,
    }


def small_batches(bench, reader, timer, library, torch, np):
    """Device-resident latency of small batches of the headline model (the kernel is chosen by batch size), each as the
    median of bursts of back-to-back launches; and four 100 000-row batches in ONE launch (memb_hip_decode_batches_device)."""
    result = []
    for count in (1000, 10000, 100000, 500000):
        picks = bench.batch_rows(len(reader), count, np)
        ids = torch.from_numpy(picks.view(np.int32)).cuda()
        target = torch.empty((count, reader.dim), dtype=torch.float32, device='cuda')
        averages = timer.bursts(lambda: reader.rows_embedding_device(ids, out=target), 100 if count <= 100000 else 30)
        ms = averages[len(averages) // 2]
        nbytes = bench.algorithmic_bytes(library, reader, picks)
        result.append({'batch': count, 'kernel': reader.info(count)['kernel'], 'lanes_per_word': reader.info(count)['lanes_per_word'],
                       'us_per_launch': ms * 1e3, 'frac': nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS})
    groups = []
    for group in range(2):   # two groups of four batches and buffers alternate: nothing cached here either
        groups.append([])
        for k in range(4):
            rows = bench.batch_rows(len(reader), 100000, np, seed=210 + 4 * group + k)
            groups[-1].append((rows, torch.from_numpy(rows.view(np.int32)).cuda(), torch.empty((100000, reader.dim), dtype=torch.float32, device='cuda')))
    turn = [0]

    def many():
        reader.rows_embedding_device_many([(ids, out) for _, ids, out in groups[turn[0] % 2]])
        turn[0] += 1

    averages = timer.bursts(many, 30)
    ms = averages[len(averages) // 2]
    nbytes = sum(bench.algorithmic_bytes(library, reader, rows) for group in groups for rows, _, _ in group) / 2
    return {'small_batches': result,
            'four_batches_of_100k_in_one_launch': {'kernel': 'decode_trained_batches', 'launch_ms': ms, 'ms_per_batch': ms / 4,
                                                   'frac': nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}}


def run(bench, reader, path, rows_host, out, timer, library, torch, np, kernel_avg_ms):
    """Everything above for the headline model, as one dictionary (bench.py --extras)."""
    extras = {}
    if reader.dim == 300 and len(rows_host) == len(reader):
        merged = torch.empty((min(500000, len(rows_host)), 600), dtype=torch.float32, device='cuda')
        ceilings = box_ceilings(torch, timer, out, len(rows_host), union=(merged, merged.shape[0]))
        del merged
        if ceilings:
            candidates = {key: ceilings[key]['ms'] for key in ('tile_fill_sequential_records', 'persistent_tile_fill_sequential_records',
                                                               'two_tiles_sequential_records') if key in ceilings}
            best = min(candidates, key=candidates.get)
            ceilings['kernel_against_the_fastest_pattern'] = {'pattern': best, 'pattern_ms': candidates[best], 'kernel_avg_ms': kernel_avg_ms,
                                                              'kernel_over_pattern': kernel_avg_ms / candidates[best]}
        extras['box_ceilings'] = ceilings
    extras.update(small_batches(bench, reader, timer, library, torch, np))
    extras['host_api'] = host_api_timings(reader, path, rows_host)
    extras['word_search'] = word_search_timings(reader, path, torch, np)
    return extras


if __name__ == '__main__':
    import sys
    sys.path.insert(0, REPO)
    sys.argv = [sys.argv[0], '--extras', '--no-configs', '--no-live-traffic'] + sys.argv[1:]
    import bench
    bench.main()
