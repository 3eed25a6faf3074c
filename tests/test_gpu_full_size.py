"""BASELINE.json's full-size configuration (GloVe-840B shape: 2,196,017 words x
300, trained 4-bit) on the GPU: the whole dump is compared with the CPU checker
slice by slice, plus size-independent properties (every value is a centroid,
permutation consistency, idempotence)."""
import os

import numpy as np
import pytest

import oracle
from conftest import bits_equal

pytestmark = pytest.mark.gpu

FULL_VOCAB = 2196017


@pytest.fixture(scope='module')
def full_model(native, tmp_path_factory):
    from memb_amd import synthetic
    count = int(os.environ.get('MEMB_TEST_FULL_VOCAB', FULL_VOCAB))
    path, _ = synthetic.cached_model(count, 300, 'trained', 4)   # shared with bench.py on the same box
    return path, count


def test_full_vocabulary_dump(native, full_model):
    import torch
    path, count = full_model
    reader = native.Reader(path)
    checker = oracle.OracleReader(path)
    assert len(reader) == count

    rows = torch.arange(count, dtype=torch.int32, device='cuda')
    out = reader.rows_embedding_device(rows)
    torch.cuda.synchronize()

    # (1) slice-by-slice bit comparison with the checker over the whole vocabulary
    step = 200000
    for start in range(0, count, step):
        stop = min(count, start + step)
        expected = checker.rows_embedding(np.arange(start, stop, dtype=np.uint32))
        assert bits_equal(out[start:stop].cpu().numpy(), expected), (start, stop)

    # (2) every decoded value is one of the <= 16 centroids
    assert torch.unique(out).numel() <= 16

    # (3) idempotence: a second launch into a fresh buffer gives the same bits
    again = reader.rows_embedding_device(rows)
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int32), again.view(torch.int32))
    del again

    # (4) permutation consistency: decode(rows[perm]) == decode(rows)[perm], misses stay zero
    generator = torch.Generator(device='cuda').manual_seed(11)
    perm = torch.randperm(count, device='cuda', generator=generator)[:500000].to(torch.int32)
    perm[::1000] = -1  # 0xFFFFFFFF
    shuffled = reader.rows_embedding_device(perm)
    torch.cuda.synchronize()
    valid = perm >= 0
    assert torch.equal(shuffled[valid].view(torch.int32), out[perm[valid].long()].view(torch.int32))
    assert not bool(shuffled[~valid].any())


def test_full_vocabulary_through_the_word_api(native, full_model):
    # reader[keys()] -- the reference's to_keyed_vectors call (python/memb/reader.py:27-28):
    # word search overlapped with decode and the pinned-ring copy, against the device-resident result
    import torch
    path, count = full_model
    reader = native.Reader(path)
    keys = reader.keys()
    assert len(keys) == count and keys == sorted(keys)

    device_rows = reader.rows_embedding_device(torch.arange(count, dtype=torch.int32, device='cuda'))
    host_rows = reader[keys]
    assert host_rows.shape == (count, 300) and host_rows.dtype == np.float32
    assert bits_equal(host_rows, device_rows.cpu().numpy())
    del host_rows

    # reversed order with unknown words mixed in, into a wider matrix (ReadersUnion's concatenation)
    sample = keys[::-7]
    sample[::100] = ['\x7fnot a word'] * len(sample[::100])
    wide = np.full((len(sample), 310), 3.0, dtype=np.float32)
    reader.batch_embedding_into(sample, wide, 10)
    expected = device_rows[torch.arange(count - 1, -1, -7, device='cuda')].cpu().numpy()
    expected[::100] = 0
    assert bits_equal(wide[:, 10:], expected)
    assert (wide[:, :10] == 3.0).all()


def test_large_batches_from_several_threads(native, full_model):
    # batches above the search/decode overlap threshold from concurrent threads on one Reader: the
    # word-search pool is taken by one caller (the others search on threads of their own), the
    # context serialises the decodes; every caller must get its own rows
    import threading
    import torch
    path, count = full_model
    reader = native.Reader(path)
    keys = reader.keys()
    expected = reader.rows_embedding_device(torch.arange(count, dtype=torch.int32, device='cuda')).cpu().numpy()
    spans = [(0, 300000), (500000, 830000), (count - 280000, count), (1000000, 1001500)]
    results = [None] * len(spans)
    errors = []

    def work(index):
        first, last = spans[index]
        try:
            for _ in range(2):
                results[index] = reader.batch_embedding(keys[first:last])
        except Exception as error:   # pragma: no cover
            errors.append(error)

    threads = [threading.Thread(target=work, args=(index,)) for index in range(len(spans))]
    for thread in threads:
        thread.start()
    for thread in threads:
        thread.join()
    assert not errors, errors
    for (first, last), rows in zip(spans, results):
        assert bits_equal(rows, expected[first:last]), (first, last)
