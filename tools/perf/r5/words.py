"""Word -> row: host search (hash index on pooled threads) against the device search (pack + one copy + one kernel),
2.2 M words in key order / shuffled and 100 000 random words, with the phases of the device path.

    python tools/perf/r5/words.py [--words N] [--repeats R]
"""
import argparse
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, REPO)


def best(call, repeats):
    times = []
    for _ in range(repeats):
        start = time.perf_counter()
        call()
        times.append(time.perf_counter() - start)
    return min(times), sorted(times)[len(times) // 2]


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument('--words', type=int, default=2196017)
    parser.add_argument('--repeats', type=int, default=7)
    args = parser.parse_args()
    import torch
    import memb_amd
    from memb_amd import synthetic, _memb
    path, seconds = synthetic.cached_model(args.words, 300, 'trained', 4, device=0)
    reader = memb_amd.Reader(path)
    start = time.perf_counter()
    keys = reader.keys()
    print('keys(): %.3f s' % (time.perf_counter() - start))
    start = time.perf_counter()
    reader.info()
    print('stage model: %.3f s' % (time.perf_counter() - start))
    start = time.perf_counter()
    reader.stage_words()
    torch.cuda.synchronize()
    info = reader.info()
    print('stage words: %.3f s, %d keys, %d slots, %.1f MB' % (
        time.perf_counter() - start, info['word_index_keys'], info['word_index_slots'], info['word_index_bytes'] / 1e6))
    rng = np.random.default_rng(11)
    order = rng.permutation(len(keys))
    batches = {
        'key order, all': keys,
        'shuffled, all': [keys[i] for i in order],
        '100 000 random': [keys[i] for i in rng.integers(0, len(keys), size=100000)],
        '10 000 random': [keys[i] for i in rng.integers(0, len(keys), size=10000)],
        '1 000 random': [keys[i] for i in rng.integers(0, len(keys), size=1000)],
    }
    batches['500 000 random'] = [keys[i] for i in rng.integers(0, len(keys), size=500000)]
    for name, words in batches.items():
        expected = reader.resolve_rows(words)
        rows = torch.empty(len(words), dtype=torch.int32, device='cuda')

        def device():
            reader.resolve_rows_device(words, out=rows)
            torch.cuda.synchronize()

        def device_no_sync():
            reader.resolve_rows_device(words, out=rows)

        device()
        assert np.array_equal(rows.cpu().numpy().view(np.uint32), expected), name
        host = best(lambda: reader.resolve_rows(words), args.repeats)
        dev = best(device, args.repeats)
        torch.cuda.synchronize()
        enqueue = best(device_no_sync, args.repeats)
        torch.cuda.synchronize()
        scratch = _memb.WordBatch(0)
        views = min(_memb._word_fill_seconds(scratch, words) for _ in range(args.repeats))
        # kernel alone: events around a second lookup of the packed batch
        batch = reader._word_batch
        begin, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        stream = torch.cuda.current_stream().cuda_stream
        kernel = []
        for _ in range(args.repeats):
            begin.record()
            reader._impl.resolve_batch_to_device(batch, rows.data_ptr(), stream)
            end.record()
            torch.cuda.synchronize()
            kernel.append(begin.elapsed_time(end))
        print('%-18s n=%8d  host %.3f ms (median %.3f) | device %.3f ms (median %.3f) = %.1fx | host side of the device call %.3f ms, fill alone (no lookups) %.3f ms | kernel %.3f ms' % (
            name, len(words), host[0] * 1e3, host[1] * 1e3, dev[0] * 1e3, dev[1] * 1e3, host[0] / dev[0], enqueue[0] * 1e3, views * 1e3, min(kernel)))


if __name__ == '__main__':
    main()
