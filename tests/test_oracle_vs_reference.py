"""Randomised comparison of the restated decoder with the reference's own
headers (oracle/_ref). Runs only where oracle/_ref was built, i.e. where
/root/reference exists or the prebuilt library travelled with the tree."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.skipif(not oracle.reference_available(), reason='oracle/_ref not built')


def random_code(rng, symbols):
    """random complete prefix code: split leaves until `symbols` leaves exist"""
    leaves = [0]
    while len(leaves) < symbols:
        i = int(rng.integers(0, len(leaves)))
        if leaves[i] >= 16:
            candidates = [j for j, depth in enumerate(leaves) if depth < 16]
            i = candidates[int(rng.integers(0, len(candidates)))]
        depth = leaves.pop(i)
        leaves += [depth + 1, depth + 1]
    lengths = sorted(leaves)
    keys = rng.permutation(255)[:symbols].astype(np.uint8)
    size_offsets = []
    current = 0
    for i, length in enumerate(lengths):
        while current < length:
            current += 1
            size_offsets.append(i)
    size_offsets.append(symbols)
    return keys, lengths, size_offsets


def test_decoders_agree_on_random_codes():
    rng = np.random.default_rng(77)
    ours = oracle.Codec('oracle')
    reference = oracle.Codec('reference')
    for trial in range(150):
        symbols = int(rng.integers(2, 200))
        keys, lengths, size_offsets = random_code(rng, symbols)
        codes_a = ours.canonical_codes(keys, lengths)
        codes_b = reference.canonical_codes(keys, lengths)
        assert np.array_equal(codes_a[0], codes_b[0]) and np.array_equal(codes_a[1], codes_b[1])
        message = keys[rng.integers(0, symbols, size=int(rng.integers(1, 400)))]
        stream_a = ours.bitstream_pack(codes_a[0][message], codes_a[1][message])
        stream_b = reference.bitstream_pack(codes_b[0][message], codes_b[1][message])
        assert np.array_equal(stream_a, stream_b)
        for bits in (1, 2, int(rng.integers(3, 13)), 10):
            count = len(message) + int(rng.integers(0, 9))  # the tail reads zero-filled bits past the end
            got = ours.decode_symbols(keys, size_offsets, bits, stream_a, count)
            want = reference.decode_symbols(keys, size_offsets, bits, stream_a, count)
            assert np.array_equal(got, want), (trial, bits)
            assert np.array_equal(got[:len(message)], message)


@pytest.mark.parametrize('bits,distribution', [(2, 'normal'), (4, 'normal'), (6, 'student'), (8, 'student')])
def test_reference_decoder_rows_equal_restatement_on_whole_files(make_model, bits, distribution):
    # whole-file check: the reference's HuffmanTableDecoder + centroid gather over every row of a
    # synthetic model (written by memb_amd.Builder) against the C restatement's Reader
    path, words = make_model(3000, 300, 'trained', bits, distribution=distribution)
    reader = oracle.OracleReader(path)
    rows = np.concatenate([np.arange(len(words), dtype=np.uint32)[::-1],
                           np.array([0xFFFFFFFF, len(words), 0], dtype=np.uint32)])
    expected = reader.rows_embedding(rows)
    assert not expected[-3:-1].any()
    for max_direct_bits in (10, 1, 3):   # 10 = reference default (src/trained_compression.h:11), 1 = src/tests.cpp:76-88
        decoder = oracle.ReferenceDecoder(reader, max_direct_bits)
        for threads in (1, 3):
            got = decoder.rows_embedding(rows, num_threads=threads)
            assert np.array_equal(got.view(np.uint32), expected.view(np.uint32)), (max_direct_bits, threads)
        decoder.close()


def test_reference_decoder_on_the_golden_models():
    import json
    import os
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, 'models.json')) as handle:
        entries = json.load(handle)
    for entry in entries:
        if entry['storage'] != 'trained':
            continue
        reader = oracle.OracleReader(os.path.join(GOLDEN, entry['file']))
        rows = np.arange(len(entry['keys']), dtype=np.uint32)
        got = oracle.ReferenceDecoder(reader).rows_embedding(rows)
        assert np.array_equal(got.view(np.uint32), np.load(os.path.join(GOLDEN, entry['rows'])).view(np.uint32))
