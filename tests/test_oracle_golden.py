"""The CPU checker (oracle/memb_oracle.c) against golden vectors produced by the
reference's own decode-side code (tests/golden/make_golden.py) and against the
reference's own test cases."""
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, SIX_WORDS, bits_equal, golden_json


def test_huffman_table_decoder_matches_reference_vectors():
    # reference src/huffman_table_decoder.h:102-118 + src/bit_stream_reader.h:16-31
    codec = oracle.Codec('oracle')
    cases = golden_json('huffman_decode.json')
    assert len(cases) > 40
    for case in cases:
        stream = np.frombuffer(bytes.fromhex(case['stream']), dtype=np.uint8)
        expected = np.frombuffer(bytes.fromhex(case['symbols']), dtype=np.uint8)
        got = codec.decode_symbols(case['keys'], case['size_offsets'], case['max_direct_bits'], stream, case['count'])
        assert np.array_equal(got, expected), case['name']


def test_canonical_prefix_codes_match_reference_vectors():
    # reference src/prefix_code.cpp:5-22
    codec = oracle.Codec('oracle')
    for case in golden_json('canonical_codes.json'):
        codes, bits = codec.canonical_codes(case['keys'], case['lengths'])
        assert [int(codes[k]) for k in case['keys']] == case['codes']
        assert [int(bits[k]) for k in case['keys']] == case['bits']


def test_bit_stream_known_answer():
    # reference src/bit_stream_tests.cpp:31-59
    known = golden_json('bit_stream.json')
    packed = oracle.Codec('oracle').bitstream_pack([c for c, _ in known['codes']], [n for _, n in known['codes']])
    assert packed.tobytes().hex() == known['bytes']
    assert ''.join(format(x, '08b') for x in packed) == known['bit_string']


@pytest.mark.parametrize('max_direct_bits', [0, 1, 3, 12])
def test_model_files_decode_to_reference_rows(max_direct_bits):
    # reference src/trained_compression.cpp:113-140; L = 1 is the reference's
    # indirect-table test (src/tests.cpp:76-88)
    for entry in golden_json('models.json'):
        if entry['storage'] != 'trained':
            continue
        reader = oracle.OracleReader(os.path.join(GOLDEN, entry['file']), 1, max_direct_bits)
        assert reader.keys() == entry['keys']
        rows = np.load(os.path.join(GOLDEN, entry['rows']))
        assert bits_equal(reader.batch_embedding(entry['keys']), rows), entry['file']
        assert bits_equal(reader.rows_embedding(np.arange(len(rows), dtype=np.uint32)), rows)


@pytest.mark.parametrize('storage', ['full', 'uniform', 'trained'])
def test_builder_round_trip_like_reference(storage):
    # reference src/tests.cpp:29-74: sorted keys, 1 % tolerance, missing word -> zeros
    reader = oracle.OracleReader(os.path.join(GOLDEN, 'six_words_{}.bin'.format(storage)))
    assert reader.dim == 3
    assert reader.keys() == sorted(SIX_WORDS)
    for word, vector in SIX_WORDS.items():
        embedding = reader.word_embedding(word)
        assert embedding.shape == (3,)
        for got, want in zip(embedding, vector):
            assert abs(got - want) <= 0.01 * max(abs(got), abs(want)) or (want == 0 and abs(got) < 1e-6)
    assert (reader.word_embedding('o') == 0.0).all()
    assert (reader.word_embedding('zzz') == 0.0).all()  # sorts after every key: end() in the reference


def test_threaded_decoder_equals_serial():
    # reference src/tests.cpp:90-113: 1025 words, 1 thread vs 4 threads, bitwise equal
    path = os.path.join(GOLDEN, 'six_words_trained.bin')
    words = list(SIX_WORDS)
    batch = [words[i % len(words)] for i in range(1025)]
    serial = oracle.OracleReader(path, 1).batch_embedding(batch)
    threaded = oracle.OracleReader(path, 4).batch_embedding(batch)
    assert bits_equal(serial, threaded)


def test_missing_and_invalid_files(tmp_path):
    # reference src/tests.cpp:134-153
    with pytest.raises(RuntimeError):
        oracle.OracleReader(str(tmp_path / 'missing.bin'))
    invalid = tmp_path / 'invalid.bin'
    invalid.write_bytes(b'0123456789')
    with pytest.raises(RuntimeError, match='File format verification failed'):
        oracle.OracleReader(str(invalid))


def test_uniform_expression_is_ieee_fp32():
    # reference src/uniform_compression.cpp:70-71. The vectors come from oracle/uniform_expr.cpp (the
    # expression as a C++ translation unit built with the reference's flags); the C restatement, that
    # translation unit as built here, and numpy's float32 arithmetic must all reproduce them -- levels 0
    # (absent field: +-inf, NaN), max < min and subnormal rows included. NaNs compare as NaNs.
    cases = golden_json('uniform_expr.json')
    assert {c[3] for c in cases} == {0, 1, 2, 16, 255}
    assert any(np.uint32(c[0]).view(np.float32) > np.uint32(c[1]).view(np.float32) for c in cases)   # max < min
    subnormal = [c for c in cases if 0 < (c[0] & 0x7fffffff) < 0x00800000 and 0 < (c[1] & 0x7fffffff) < 0x00800000]
    assert len(subnormal) > 50 and any(0 < (c[4] & 0x7fffffff) < 0x00800000 for c in subnormal)
    assert any(np.isinf(np.uint32(c[4]).view(np.float32)) for c in cases if c[3] == 0)
    assert any(np.isnan(np.uint32(c[4]).view(np.float32)) for c in cases if c[3] == 0)
    mismatches = 0
    with np.errstate(all='ignore'):
        for low, high, value, levels, expected in cases:
            low = np.uint32(low).view(np.float32)
            high = np.uint32(high).view(np.float32)
            want = np.uint32(expected).view(np.float32)
            candidates = (
                np.float32(oracle.uniform_value(low, high, value, levels)),
                oracle.uniform_expression(low, high, levels, [value])[0],
                np.float32(low + (high - low) * np.float32(value) / np.float32(levels)),
            )
            for got in candidates:
                if np.isnan(want):
                    mismatches += not np.isnan(got)
                else:
                    mismatches += got.view(np.uint32) != np.uint32(expected)
    assert mismatches == 0


def test_uniform_file_matches_expression():
    reader = oracle.OracleReader(os.path.join(GOLDEN, 'six_words_uniform.bin'))
    # 'the' = [0, 1, 2]: min 0, max 2, levels 255 -> bytes 0, 127, 255
    assert bits_equal(reader.word_embedding('the'),
                      [oracle.uniform_value(0.0, 2.0, v, 255) for v in (0, 127, 255)])
