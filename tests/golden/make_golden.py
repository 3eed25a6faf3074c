"""Generate the golden vectors under tests/golden/.

Run in the build container, where /root/reference exists:

    python build_native.py && python tests/golden/make_golden.py

Expected outputs come from the REFERENCE's own decode-side code -- its
std-only headers huffman_table_decoder.h / bit_stream_reader.h / bit_stream.h
and prefix_code.cpp, compiled in place into oracle/_ref/libmemb_ref.so
(oracle/Makefile) -- never from the oracle restatement or the HIP path:

  huffman_decode.json   (keys, size_offsets, L, stream) -> symbols from
                        HuffmanTableDecoder::next, incl. reads past the end
  canonical_codes.json  createCanonicalPrefixCodes
  bit_stream.json       BitStream::push on the reference's known-answer input
                        (src/bit_stream_tests.cpp:35-41)
  *.bin + *.rows.npy    small model files written by memb_amd.Builder and the
                        rows obtained by parsing the file here (own FlatBuffers
                        walk below), decoding every stream with the reference
                        decoder and gathering centroids with numpy
  uniform_expr.json     NOT reference output (uniform_compression.cpp needs
                        flatc-generated headers): the expression at
                        src/uniform_compression.cpp:70-71 evaluated with
                        numpy float32 scalars, an independent IEEE evaluation
"""
import heapq
import json
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle  # noqa: E402
import memb_amd  # noqa: E402
from memb_amd import synthetic  # noqa: E402

REF = oracle.Codec('reference')

SIX_WORDS = {  # reference src/tests.cpp:20-27
    'the': [0.0, 1.0, 2.0], 'of': [0.0, -1.0, 2.0], 'th': [2.0, 0.0, 1.0],
    'a': [1.0, 0.0, -2.0], 'tho': [2.0, 0.0, -1.0], 'abc': [-2.0, 0.0, 1.0],
}


def huffman_lengths(counts):
    """code length per symbol for positive counts (any optimal tie-breaking)"""
    heap = [(c, i, (k,)) for i, (k, c) in enumerate(sorted(counts.items()))]
    heapq.heapify(heap)
    lengths = {k: 0 for k in counts}
    serial = len(heap)
    while len(heap) > 1:
        c1, _, s1 = heapq.heappop(heap)
        c2, _, s2 = heapq.heappop(heap)
        for k in s1 + s2:
            lengths[k] += 1
        heapq.heappush(heap, (c1 + c2, serial, s1 + s2))
        serial += 1
    return lengths


def decoder_description(lengths):
    """keys by increasing length + size_offsets (reference src/huffman_encoder.cpp:100-117)"""
    ordered = sorted(lengths.items(), key=lambda item: (item[1], item[0]))
    keys = [k for k, _ in ordered]
    size_offsets = []
    current = 0
    for i, (_, length) in enumerate(ordered):
        while current < length:
            current += 1
            size_offsets.append(i)
    size_offsets.append(len(ordered))
    return keys, [l for _, l in ordered], size_offsets


def encode(keys, lengths, symbols):
    codes, bits = REF.canonical_codes(keys, lengths)
    return REF.bitstream_pack(codes[symbols], bits[symbols])


def decode_cases():
    rng = np.random.default_rng(20240)
    cases = []

    def add(name, counts, count, table_bits, tail_symbols=0):
        lengths = huffman_lengths(counts)
        if max(lengths.values()) > 16:
            return
        keys, sorted_lengths, size_offsets = decoder_description(lengths)
        alphabet = np.array(sorted(counts), dtype=np.uint8)
        weights = np.array([counts[k] for k in sorted(counts)], dtype=np.float64)
        symbols = rng.choice(alphabet, size=count, p=weights / weights.sum())
        # make sure the rarest (longest-code) symbols occur
        rare = np.array(keys[-min(len(keys), 8):], dtype=np.uint8)
        symbols[rng.integers(0, count, size=len(rare))] = rare
        stream = encode(keys, sorted_lengths, symbols)
        for bits in table_bits:
            total = count + tail_symbols  # tail: decode past the end, zero-filled
            expected = REF.decode_symbols(keys, size_offsets, bits, stream, total)
            assert (expected[:count] == symbols).all(), name
            cases.append({
                'name': '{}-L{}'.format(name, bits), 'keys': keys, 'size_offsets': size_offsets,
                'max_direct_bits': bits, 'stream': stream.tobytes().hex(), 'count': total,
                'symbols': expected.tobytes().hex(),
            })

    for levels, spread in ((4, 1.0), (16, 2.2), (41, 3.0), (172, 3.3), (255, 3.6)):
        centers = np.linspace(-spread, spread, levels)
        weights = np.exp(-0.5 * centers ** 2) + 1e-5
        counts = {int(k): max(1, int(w * 1e6)) for k, w in enumerate(weights)}
        add('gauss{}'.format(levels), counts, 300, (1, 2, 3, 5, 8, 10, 12), tail_symbols=7)
    fib = [1, 1]
    while len(fib) < 17:
        fib.append(fib[-1] + fib[-2])
    add('fibonacci17', {k: c for k, c in enumerate(fib)}, 400, (1, 4, 10, 12), tail_symbols=5)  # max length 16
    add('two', {3: 5, 9: 1}, 64, (1, 10))
    add('uniform8', {k: 1 for k in range(8)}, 128, (1, 2, 3, 10))

    # single symbol: zero-length code, empty stream (reference src/huffman_table_decoder.h:44-56)
    keys, size_offsets = [7], [1]
    for bits in (1, 10):
        expected = REF.decode_symbols(keys, size_offsets, bits, np.zeros(0, dtype=np.uint8), 20)
        cases.append({'name': 'single-L{}'.format(bits), 'keys': keys, 'size_offsets': size_offsets,
                      'max_direct_bits': bits, 'stream': '', 'count': 20, 'symbols': expected.tobytes().hex()})
    return cases


def canonical_cases():
    cases = []
    for lengths in ([1, 2, 3, 3], [3] * 6 + [4] * 2 + [5] * 3 + [6, 7, 8, 9, 9], [0], [16] * 4 + [15, 14, 2, 1][::-1]):
        lengths = sorted(lengths)
        keys = list(range(40, 40 + len(lengths)))
        codes, bits = REF.canonical_codes(keys, lengths)
        cases.append({'keys': keys, 'lengths': lengths,
                      'codes': [int(codes[k]) for k in keys], 'bits': [int(bits[k]) for k in keys]})
    return cases


# ---- FlatBuffers walk (FlatBuffers binary spec), independent of the C++ and C parsers ----

def _u32(b, p):
    return struct.unpack_from('<I', b, p)[0]


def _field(b, table, field_id):
    vtable = table - struct.unpack_from('<i', b, table)[0]
    if 4 + 2 * field_id + 2 > struct.unpack_from('<H', b, vtable)[0]:
        return 0
    offset = struct.unpack_from('<H', b, vtable + 4 + 2 * field_id)[0]
    return table + offset if offset else 0


def _indirect(b, table, field_id):
    p = _field(b, table, field_id)
    return p + _u32(b, p)


def _vector(b, table, field_id, dtype):
    p = _indirect(b, table, field_id)
    return np.frombuffer(b, dtype=dtype, count=_u32(b, p), offset=p + 4)


def rows_via_reference_decoder(path, max_direct_bits):
    """Sorted keys and rows of a trained file: reference decoder + numpy gather."""
    b = open(path, 'rb').read()
    assert b[4:8] == b'memb'
    index = _u32(b, 0)
    assert b[_field(b, index, 0)] == 3
    dim = _u32(b, _field(b, index, 2))
    storage = _indirect(b, index, 1)
    word_offsets = _vector(b, storage, 0, np.uint32)
    value_offsets = _vector(b, storage, 1, np.uint32)
    packed_words = _vector(b, storage, 2, np.uint8).tobytes()
    packed_values = _vector(b, storage, 3, np.uint8)
    decoder = _indirect(b, storage, 4)
    keys = _vector(b, decoder, 0, np.uint8)
    size_offsets = _vector(b, decoder, 1, np.uint32)
    centroids = _vector(b, _indirect(b, storage, 5), 0, np.float32)
    words = [packed_words[o:packed_words.index(b'\0', o)].decode() for o in word_offsets]
    rows = np.empty((len(words), dim), dtype=np.float32)
    for r, offset in enumerate(value_offsets):
        symbols = REF.decode_symbols(keys, size_offsets, max_direct_bits, packed_values[offset:], dim)
        rows[r] = centroids[symbols]
    return words, rows


def model_files():
    manifest = []
    for storage in ('full', 'uniform', 'trained'):
        builder = memb_amd.Builder(3, storage, 8)
        for word, vector in SIX_WORDS.items():
            builder.add_word(word, np.array(vector, dtype=np.float32))
        name = 'six_words_{}.bin'.format(storage)
        builder.save(os.path.join(HERE, name))
        manifest.append({'file': name, 'storage': storage, 'dim': 3, 'bits': 8})
    for bits, count, distribution in ((2, 48, 'normal'), (4, 96, 'normal'), (6, 64, 'student'), (8, 64, 'normal')):
        name = 'synthetic_{}bit.bin'.format(bits)
        synthetic.build_file(os.path.join(HERE, name), count, 300, 'trained', bits, seed=99 + bits,
                             word_seed=5, distribution=distribution)
        manifest.append({'file': name, 'storage': 'trained', 'dim': 300, 'bits': bits})
    for entry in manifest:
        if entry['storage'] != 'trained':
            continue
        path = os.path.join(HERE, entry['file'])
        words, rows = rows_via_reference_decoder(path, 10)
        for bits in (1, 3):  # the reference's result must not depend on its table size
            assert np.array_equal(rows_via_reference_decoder(path, bits)[1], rows)
        np.save(path[:-4] + '.rows.npy', rows)
        entry['keys'] = words
        entry['rows'] = entry['file'][:-4] + '.rows.npy'
    return manifest


def uniform_expression_cases():
    """[min bits, max bits, value, levels, result bits] from oracle/uniform_expr.cpp -- the reference's
    expression (src/uniform_compression.cpp:70-71) compiled with the reference's flags. Covers the
    stored-levels values of real files (2, 16, 255), an absent `quantization_levels` field (0: division
    by zero, +-inf and NaN results), max < min, all-subnormal rows and overflowing ranges. numpy's
    float32 arithmetic must agree (NaNs compare as NaNs: sign and payload of a generated NaN are the
    platform's, 0xFFC00000 on x86-64)."""
    rng = np.random.default_rng(5)
    cases = []
    f32 = np.float32
    specials = [
        (0.0, 0.0), (-1.0, 1.0), (-2.5, 7.25), (0.1, 0.1000001), (-3.4e38, 3.4e38),
        # subnormal rows: everything below 2^-126
        (1e-40, 3e-39), (1.4e-45, 1.4e-44), (-5e-41, 5e-41), (0.0, 1.4e-45), (-1.1754942e-38, 1.1754942e-38),
        (1.1754942e-38, 1.17549435e-38),     # largest subnormal .. smallest normal
        # max < min (a range that runs backwards)
        (1.0, -1.0), (7.25, -2.5), (3e-39, 1e-40), (3.4e38, -3.4e38),
        # infinities in the file
        (0.0, float('inf')), (float('-inf'), float('inf')), (float('-inf'), 0.0),
    ]
    pairs = specials + [tuple(np.sort(rng.standard_normal(2).astype(np.float32) * 2)) for _ in range(46)]
    with np.errstate(all='ignore'):
        for low, high in pairs:
            low, high = f32(low), f32(high)
            for levels in (0, 1, 2, 16, 255):
                values = sorted({v for v in (0, 1, levels // 2, levels - 1, levels, 255) if 0 <= v <= 255})
                results = oracle.uniform_expression(low, high, levels, values)
                port = [oracle.uniform_value(low, high, v, levels) for v in values]
                for value, result, ported in zip(values, results, port):
                    viaNumpy = low + (high - low) * f32(value) / f32(levels)
                    for other in (f32(viaNumpy), f32(ported)):   # numpy float32 and oracle/memb_oracle.c agree
                        assert (np.isnan(other) and np.isnan(result)) or other.view(np.uint32) == result.view(np.uint32), \
                            (low, high, value, levels)
                    cases.append([int(low.view(np.uint32)), int(high.view(np.uint32)), int(value), levels,
                                  int(f32(result).view(np.uint32))])
    return cases


def main():
    if not oracle.reference_available():
        raise SystemExit('oracle/_ref/libmemb_ref.so missing: run `python build_native.py` where /root/reference exists')

    def dump(name, payload):
        with open(os.path.join(HERE, name), 'w') as f:
            json.dump(payload, f, separators=(',', ':'))
            f.write('\n')

    if '--only-uniform' in sys.argv:   # the other vectors stay as committed
        dump('uniform_expr.json', uniform_expression_cases())
        return
    dump('huffman_decode.json', decode_cases())
    dump('canonical_codes.json', canonical_cases())
    known = [(1023, 14), (33, 6), (0, 4), (1234, 11), (7, 2)]  # reference src/bit_stream_tests.cpp:35-41
    packed = REF.bitstream_pack([c for c, _ in known], [n for _, n in known])
    bit_string = ''.join(format(c & ((1 << n) - 1), '0{}b'.format(n)) for c, n in known)  # prettyBitString (:8-18)
    bit_string += '0' * (8 - len(bit_string) % 8)  # the test's own expectation (:44-49)
    assert ''.join(format(x, '08b') for x in packed) == bit_string
    dump('bit_stream.json', {'codes': known, 'bytes': packed.tobytes().hex(), 'bit_string': bit_string})
    dump('models.json', model_files())
    dump('uniform_expr.json', uniform_expression_cases())
    print('golden vectors written to', HERE)


if __name__ == '__main__':
    main()
