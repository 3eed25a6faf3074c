"""Host side of the product on CPU: container format, Builder, word search,
Python surface. Lookups themselves need the GPU and are in test_gpu_*.py."""
import os
import re

import numpy as np
import pytest

import oracle
from conftest import REPO, GOLDEN, SIX_WORDS, golden_json


def test_strategies_listed_like_reference(native):
    # reference src/compression_strategy.cpp:13-18,63-78
    assert native.available_compression_strategies() == ['full', 'uniform', 'trained']


def test_unknown_strategy_message(native):
    # reference src/compression_strategy.cpp:11,55-58
    with pytest.raises(RuntimeError, match='Storage strategy bogus is not supported'):
        native.Builder(3, 'bogus', 4)


def test_builder_rejects_wrong_dimension_and_duplicates(native):
    # reference src/tests.cpp:115-132, messages src/builder.cpp:12-16
    builder = native.Builder(15, 'full', 8)
    with pytest.raises(RuntimeError, match=re.escape("Vector dimension (3) for word the doesn't match builder dimension (15)")):
        builder.add_word('the', np.array([0.0, 1.0, 2.0], dtype=np.float32))
    builder = native.Builder(3, 'full', 8)
    builder.add_word('the', np.array([0.0, 1.0, 2.0], dtype=np.float32))
    with pytest.raises(RuntimeError, match='Attempt to add duplicate word the to index'):
        builder.add_word('the', np.array([2.0, 1.0, 2.0], dtype=np.float32))
    with pytest.raises(RuntimeError, match='Word vector must be 1-dimensional'):
        builder.add_word('x', np.zeros((1, 3), dtype=np.float32))  # reference python/memb_bindings.cpp:19-21


def test_missing_and_invalid_files(native, tmp_path):
    # reference src/tests.cpp:134-153
    with pytest.raises(RuntimeError):
        native.Reader(str(tmp_path / 'missing.bin'))
    invalid = tmp_path / 'invalid.bin'
    invalid.write_bytes(b'0123456789')
    with pytest.raises(RuntimeError, match='File format verification failed'):
        native.Reader(invalid)  # pathlib.Path accepted, reference python/memb/reader.py:66
    truncated = tmp_path / 'truncated.bin'
    truncated.write_bytes(open(os.path.join(GOLDEN, 'six_words_trained.bin'), 'rb').read()[:60])
    with pytest.raises(RuntimeError, match='File format verification failed'):
        native.Reader(truncated)


@pytest.mark.parametrize('storage', ['full', 'uniform', 'trained'])
def test_keys_dim_and_rows_agree_with_checker(native, storage):
    path = os.path.join(GOLDEN, 'six_words_{}.bin'.format(storage))
    reader = native.Reader(path)
    checker = oracle.OracleReader(path)
    assert reader.dim == checker.dim == 3
    assert len(reader) == 6
    assert reader.keys() == checker.keys() == sorted(SIX_WORDS)  # reference src/tests.cpp:38-43
    probes = sorted(SIX_WORDS) + ['o', '', 'zzzz', 'ab', 'abcd', 'thee', 'été']
    assert np.array_equal(reader.resolve_rows(probes), checker.resolve_rows(probes))


def test_word_search_on_a_larger_vocabulary(native, make_model):
    path, words = make_model(5000, dim=8, storage='trained', bits=4)
    reader = native.Reader(path, num_threads=4)
    checker = oracle.OracleReader(path)
    assert reader.keys() == sorted(words) == checker.keys()
    rng = np.random.default_rng(3)
    probes = [words[i] for i in rng.integers(0, len(words), size=3000)]
    probes += [w + 'x' for w in probes[:200]] + [w[:-1] for w in probes[:200]] + ['', '~~~~', '0']
    rows = reader.resolve_rows(probes)  # >= 1024 words: threaded search, reference src/reader.cpp:61-84
    assert np.array_equal(rows, checker.resolve_rows(probes))
    serial = native.Reader(path, num_threads=1).resolve_rows(probes)
    assert np.array_equal(rows, serial)
    order = {w: i for i, w in enumerate(sorted(words))}
    assert all(rows[i] == order[probes[i]] for i in range(3000))


def test_hash_index_lookup_equals_binary_search(native, make_model):
    # batches >= 4096 words go through the lazily built hash index; it must give the binary search's answers
    path, words = make_model(5000, dim=8, storage='trained', bits=4)
    checker = oracle.OracleReader(path)
    rng = np.random.default_rng(9)
    probes = [words[i] for i in rng.integers(0, len(words), size=20000)]
    probes[::7] = [w + '!' for w in probes[::7]]          # misses that share long prefixes with keys
    probes[5::11] = [w[:max(1, len(w) // 2)] for w in probes[5::11]]
    probes += ['', ' ', 'é', '日本語', words[0] + '\0tail']  # an embedded NUL ends the word, as for strcmp
    expected = checker.resolve_rows([p.split('\0')[0] for p in probes])
    for threads in (1, 3, 0):
        reader = native.Reader(path, num_threads=threads)
        small = reader.resolve_rows(probes[:100])             # binary search path (index not built yet)
        assert np.array_equal(small, expected[:100])
        assert np.array_equal(reader.resolve_rows(probes), expected)          # builds + uses the index
        assert np.array_equal(reader.resolve_rows(tuple(probes[:100])), expected[:100])  # index reused; any sequence
    # a stream of small batches gets the index too, once it has looked up as many words as a batch that builds it
    reader = native.Reader(path)
    assert not reader._impl.has_word_index()
    for start in range(0, 4000, 500):
        assert np.array_equal(reader.resolve_rows(probes[start:start + 500]), expected[start:start + 500])
    assert not reader._impl.has_word_index()
    assert np.array_equal(reader.resolve_rows(probes[4000:4500]), expected[4000:4500])
    assert reader._impl.has_word_index()
    assert np.array_equal(reader.resolve_rows(probes[:300]), expected[:300])
    with pytest.raises(TypeError):
        native.Reader(path).resolve_rows(['ok', 3])
    for storage in ('uniform', 'full'):
        path, words = make_model(4500, dim=4, storage=storage, bits=8)
        probes = sorted(words) + ['nope', 'zzzzzz', '']
        assert np.array_equal(native.Reader(path).resolve_rows(probes), oracle.OracleReader(path).resolve_rows(probes))


def test_builder_files_are_read_by_the_checker(native, tmp_path):
    # container written by memb_amd.Builder, parsed by an independent C parser
    for storage in ('full', 'uniform', 'trained'):
        builder = native.Builder(3, storage, 8)
        for word, vector in SIX_WORDS.items():
            builder.add_word(word, np.array(vector, dtype=np.float32))
        path = tmp_path / (storage + '.bin')
        builder.save(path)
        assert path.read_bytes() == open(os.path.join(GOLDEN, 'six_words_{}.bin'.format(storage)), 'rb').read()
        checker = oracle.OracleReader(str(path))
        assert checker.keys() == sorted(SIX_WORDS)
        if storage == 'full':
            for word, vector in SIX_WORDS.items():
                assert checker.word_embedding(word).tolist() == vector


def test_kmeans_known_answer(native):
    # reference src/kmeans_tests.cpp:9-38
    data = [-0.5 + i * 0.125 for i in range(8)] + [-9 + i * 0.125 for i in range(16)] + [11.75 + i * 0.125 for i in range(4)]
    centroids, assignments = native._memb._kmeans_fit_predict(data, 3)
    assert assignments == [1] * 8 + [0] * 16 + [2] * 4
    assert centroids == sorted(centroids) and len(centroids) == 3


def test_bit_packer_known_answer(native):
    # reference src/bit_stream_tests.cpp:31-59
    known = golden_json('bit_stream.json')
    assert native._memb._bit_pack([tuple(c) for c in known['codes']]).hex() == known['bytes']


def test_host_bit_packer_against_the_reference_bitstream(native):
    """The host writer's packer against the reference's own BitStream::push (src/bit_stream.h:18-34, compiled into
    oracle/_ref) on canonical codes the reference's createCanonicalPrefixCodes (src/prefix_code.cpp) assigned."""
    if not oracle.reference_available():
        pytest.skip('oracle/_ref is not built')
    reference = oracle.Codec('reference')
    rng = np.random.default_rng(23)
    for trial in range(30):
        counts = [0] * 256
        for key in rng.permutation(255)[:int(rng.integers(2, 255))]:
            counts[int(key)] = int(rng.integers(1, 10 ** int(rng.integers(1, 7))))
        keys, size_offsets = native._memb._huffman_description(counts)
        lengths = [next(k for k, bound in enumerate(size_offsets) if i < bound) for i in range(len(keys))]
        if max(lengths) > 16:
            continue
        codes, bits = reference.canonical_codes(keys, lengths)
        message = np.array(keys, dtype=np.uint8)[rng.integers(0, len(keys), size=int(rng.integers(0, 700)))]
        ours = native._memb._bit_pack([(int(codes[s]), int(bits[s])) for s in message])
        assert bytes(ours) == reference.bitstream_pack(codes[message], bits[message]).tobytes()


def test_a_failed_device_call_disables_the_builder(native):
    """A Builder whose device cannot be reached must not go on with words registered and no rows behind them
    (a caller may catch the exception of one block and add the next: tools/converter does): every later call
    refuses, nothing is written."""
    if native.hip_device_count() > 0:
        pytest.skip('a HIP device is present')
    from memb_amd import synthetic
    words = synthetic.make_words(10500)
    vectors = synthetic.make_vectors(10500, 4, seed=1)
    builder = native.Builder(4, 'trained', 4, device=0)
    with pytest.raises(RuntimeError, match='HIP device'):
        builder.add_words(words[:10200], vectors[:10200])   # the k-means sample is complete: the encoder starts
    with pytest.raises(RuntimeError, match='cannot be used further'):
        builder.add_words(words[10200:], vectors[10200:])
    with pytest.raises(RuntimeError, match='cannot be used further'):
        builder.add_word('one-more', vectors[0])
    with pytest.raises(RuntimeError, match='cannot be used further'):
        builder.save(os.devnull)


def test_encoder_description_and_device_table_decode_like_reference(native):
    """Builder's Huffman description -> (a) accepted by the reference decoder,
    (b) the product's own lookup table (memb_amd/csrc/codec.h) resolves every
    code to the same symbol and length, for one- and two-level layouts."""
    rng = np.random.default_rng(11)
    checker = oracle.Codec('reference' if oracle.reference_available() else 'oracle')
    for trial in range(40):
        symbols = int(rng.integers(1, 255))
        counts = [0] * 256
        for key in rng.permutation(255)[:symbols]:
            counts[int(key)] = int(rng.integers(1, 10 ** int(rng.integers(1, 7))))
        keys, size_offsets = native._memb._huffman_description(counts)
        lengths = []
        for i in range(len(keys)):
            lengths.append(next(k for k, bound in enumerate(size_offsets) if i < bound))
        if max(lengths) > 16:
            continue
        codes, bits = checker.canonical_codes(keys, lengths)
        message = np.array(keys, dtype=np.uint8)[rng.integers(0, len(keys), size=200)]
        stream = checker.bitstream_pack(codes[message], bits[message])
        assert np.array_equal(checker.decode_symbols(keys, size_offsets, 10, stream, 200), message)
        for limit in (1, 4, 11, 12):
            root_bits, max_bits, has_sub, table = native._memb._decode_table(keys, size_offsets, limit)
            assert max_bits == max(lengths) and root_bits == max(1, min(limit, max_bits))
            assert has_sub == (max_bits > root_bits)
            for key, length in zip(keys, lengths):
                code = int(codes[key])
                padded = code << (16 - length) if length <= 16 else 0
                entry = table[padded >> (16 - root_bits)]
                if entry & 0x80000000:
                    sub_bits = entry & 0xff
                    base = (entry & 0x7fffffff) >> 8
                    entry = table[base + ((padded >> (16 - root_bits - sub_bits)) & ((1 << sub_bits) - 1))]
                assert entry & 0xff == length and (entry >> 8) & 0xff == key


def test_lookup_without_device_fails_loudly(native):
    if native.hip_device_count() > 0:
        pytest.skip('a HIP device is present')
    reader = native.Reader(os.path.join(GOLDEN, 'six_words_trained.bin'))
    with pytest.raises(RuntimeError, match='no HIP device available'):
        reader['the']
    with pytest.raises(RuntimeError, match='no HIP device available'):
        reader[['the', 'of']]


def test_strings_without_their_terminator_are_refused(native, tmp_path):
    # storages keep raw char pointers into the mapping and strcmp them: a word whose NUL has been
    # overwritten must be refused when the file is opened, not read past
    for name, word in (('six_words_uniform.bin', b'tho'), ('six_words_full.bin', b'abc')):
        data = bytearray(open(os.path.join(GOLDEN, name), 'rb').read())
        at = data.find(len(word).to_bytes(4, 'little') + word + b'\0')
        assert at >= 0
        data[at + 4 + len(word)] = ord('x')
        broken = tmp_path / name
        broken.write_bytes(bytes(data))
        with pytest.raises(RuntimeError, match='File format verification failed'):
            native.Reader(str(broken))
    data = bytearray(open(os.path.join(GOLDEN, 'six_words_trained.bin'), 'rb').read())
    packed = b'a\0abc\0of\0th\0the\0tho\0'   # every word with its NUL; the string's own terminator follows
    at = data.find(len(packed).to_bytes(4, 'little') + packed + b'\0')
    assert at >= 0
    data[at + 4 + len(packed)] = ord('x')
    broken = tmp_path / 'trained.bin'
    broken.write_bytes(bytes(data))
    with pytest.raises(RuntimeError, match='File format verification failed'):
        native.Reader(str(broken))


def test_into_calls_refuse_arrays_they_could_not_fill_in_place(native):
    # pybind11 would convert a float64 / float16 / non-contiguous array into a temporary copy, fill
    # that, and leave the caller's matrix untouched: such arrays are refused before any lookup
    reader = native.Reader(os.path.join(GOLDEN, 'six_words_trained.bin'))
    words = ['the', 'of']
    rows = np.zeros(2, dtype=np.uint32)
    good = np.zeros((2, 6), dtype=np.float32)
    bad = [
        np.zeros((2, 6), dtype=np.float64),
        np.zeros((2, 6), dtype=np.float16),
        np.zeros((2, 6), dtype=np.int32),
        np.zeros((6, 2), dtype=np.float32).T,          # column-major view
        good[:, ::2],                                   # strided columns
        good[::-1],                                     # negative row stride
        np.zeros(12, dtype=np.float32),                 # one-dimensional
    ]
    for out in bad:
        with pytest.raises(TypeError):
            reader.batch_embedding_into(words, out, 0)
        with pytest.raises(TypeError):
            reader.rows_embedding_into(rows, out, 0)
    frozen = np.zeros((2, 6), dtype=np.float32)
    frozen.setflags(write=False)
    with pytest.raises(TypeError, match='read-only'):
        reader.rows_embedding_into(rows, frozen, 0)
    with pytest.raises(RuntimeError, match='Output must be'):   # right kind of array, wrong shape
        reader.rows_embedding_into(rows, np.zeros((3, 6), dtype=np.float32), 0)
    with pytest.raises(RuntimeError, match='Output must be'):
        reader.batch_embedding_into(words, np.zeros((2, 6), dtype=np.float32), 4)


def test_bench_refuses_more_ranks_than_devices_without_touching_a_gpu(native):
    # `python bench.py --gpus N` with no launcher starts its own ranks; on a host with fewer devices it
    # says so instead (the parent counts devices, it never initialises one)
    import subprocess
    import sys
    if native.hip_device_count() >= 2:
        pytest.skip('this host has the devices')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MEMB_BENCH_REHEARSAL')}
    run = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--small'], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert run.returncode != 0 and '--gpus 2: this node has' in run.stderr and not run.stdout.strip()


def test_getitem_dispatch_and_tokenizer_layout(native):
    # reference python/memb/reader.py:6-17,94-111
    from memb_amd.reader import BaseReader, tokenizer_word_list

    class Echo(BaseReader):
        dim = 2

        def keys(self):
            return []

        def word_embedding(self, word):
            return ('word', word)

        def batch_embedding(self, words):
            return ('batch', words)

        def tokenizer_embedding(self, tokenizer):
            return self.batch_embedding(tokenizer_word_list(tokenizer))

    echo = Echo()
    assert echo['a'] == ('word', 'a')
    assert echo[['a', 'b']] == ('batch', ['a', 'b'])
    with pytest.raises(TypeError, match='Key type is not supported'):
        echo[('a', 'b')]

    class Tokenizer:
        word_index = {'the': 1, 'of': 2, 'rare': 5}
        num_words = None

    assert tokenizer_word_list(Tokenizer) == ['', 'the', 'of', '', '', 'rare']
    Tokenizer.num_words = 3
    assert tokenizer_word_list(Tokenizer) == ['', 'the', 'of']  # idx < num_words


def test_readers_union_argument_checks(native):
    # reference python/memb/readers_union.py:59-65,8-10
    class Fake:
        def __init__(self, dim):
            self.dim = dim

        def keys(self):
            return ['b', 'a'] if self.dim == 3 else ['c', 'a']

    with pytest.raises(AssertionError, match='at least 2 readers'):
        native.ReadersUnion([Fake(3)], 'concatenate')
    with pytest.raises(KeyError):
        native.ReadersUnion([Fake(3), Fake(3)], 'sum')
    with pytest.raises(AssertionError, match='must be equal for average mode'):
        native.ReadersUnion([Fake(3), Fake(4)], 'average')
    union = native.ReadersUnion([Fake(3), Fake(4)], 'concatenate')
    assert union.dim == 7
    assert union.keys() == ['a', 'b', 'c']
    assert native.ReadersUnion([Fake(3), Fake(3)], 'average').dim == 3


def test_writer_round_trip_properties(native, tmp_path):
    """Builder -> file -> CPU checker on random small models: every decoded trained value is a
    centroid and (up to rounding at the mid-point) the nearest one to the original weight; every
    uniform value is within one quantisation step; keys come back sorted; odd shapes included."""
    rng = np.random.default_rng(123)
    for trial in range(24):
        dim = int(rng.integers(1, 41))
        count = int(rng.integers(30, 400))
        bits = int(rng.choice([1, 2, 3, 4, 6, 8]))
        scale = float(rng.choice([0.01, 0.4, 30.0]))
        vectors = (rng.standard_t(4, size=(count, dim)) * scale).astype(np.float32)
        words = ['w{}_{}'.format(trial, i) for i in rng.permutation(count)]
        for storage in ('trained', 'uniform'):
            builder = native.Builder(dim, storage, bits)
            builder.add_words(words, vectors)
            path = tmp_path / 'm{}_{}.bin'.format(trial, storage)
            builder.save(path)
            checker = oracle.OracleReader(str(path))
            assert checker.keys() == sorted(words)
            decoded = checker.batch_embedding(words)
            if storage == 'trained':
                centroids = np.unique(decoded)
                assert len(centroids) <= min(2 ** bits, 255)
                nearest = np.abs(vectors[..., None] - centroids[None, None, :]).min(axis=-1)
                assert np.all(np.abs(vectors - decoded) <= nearest * (1 + 1e-5) + 1e-6 * scale), (trial, dim, bits)
            else:
                span = vectors.max(axis=1, keepdims=True) - vectors.min(axis=1, keepdims=True)
                step = span / min(2 ** bits, 255)
                assert np.all(np.abs(vectors - decoded) <= step * 1.0001 + 1e-6 * scale), (trial, dim, bits)


def test_product_never_touches_the_checker():
    # oracle/ is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's
    # cpu_baseline leg may use it; the shipped package and its libraries must not import,
    # link or load it, and there is no CPU decode in them to fall back on
    import re
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    package = os.path.join(repo, 'memb_amd')
    for root, _, files in os.walk(package):
        for name in files:
            if name.endswith(('.py', '.h', '.cpp', '.hip')):
                with open(os.path.join(root, name), encoding='utf-8') as handle:
                    text = handle.read()
                assert not re.search(r'^\s*(import|from)\s+oracle\b', text, re.M), name
                assert 'memb_oracle' not in text and 'libmemb_ref' not in text, name
    for library in ('libmemb_hip.so',):
        needed = subprocess.run(['readelf', '-d', os.path.join(package, library)], stdout=subprocess.PIPE, text=True).stdout
        assert 'oracle' not in needed and 'memb_ref' not in needed
    with open(os.path.join(repo, '__graft_entry__.py')) as handle:
        entry = handle.read()
    build_body = entry[entry.index('def build('):entry.index('def smoke(')]
    assert not re.search(r'^\s*(import|from)\s+oracle\b', build_body, re.M)   # building the checker is not using it


def test_files_with_omitted_default_scalars(native, tmp_path):
    # The official FlatBuffers writers omit scalar fields that equal their default (here: a uniform
    # row whose min or max is exactly 0.0; readers supply it) and share one vtable between tables of
    # the same layout (so a vtable can sit behind its table: negative soffset). Write such files and
    # read them with the C parser of the checker and the C++ reader of the package (the GPU half is
    # in tests/test_gpu_parity.py::test_omitted_default_scalars_on_device).
    from memb_amd import _memb
    vectors = {
        'low_zero': [0.0, 1.0, 2.0], 'high_zero': [-2.0, -1.0, 0.0], 'all_zero': [0.0, 0.0, 0.0], 'plain': [-1.0, 0.5, 3.0],
    }
    paths = {}
    for omit in (False, True):
        _memb._writer_mimics_official_layout(omit)
        try:
            builder = native.Builder(3, 'uniform', 8)
            for word, vector in vectors.items():
                builder.add_word(word, np.array(vector, dtype=np.float32))
            paths[omit] = str(tmp_path / 'uniform_omit_{}.bin'.format(int(omit)))
            builder.save(paths[omit])
        finally:
            _memb._writer_mimics_official_layout(False)
    plain, omitted = (open(paths[flag], 'rb').read() for flag in (False, True))
    assert len(omitted) <= len(plain) - 40   # the zero scalars and the repeated vtables really are gone
    words = sorted(vectors) + ['missing']
    reference_rows = oracle.OracleReader(paths[False]).batch_embedding(words)
    assert np.array_equal(oracle.OracleReader(paths[True]).batch_embedding(words).view(np.uint32), reference_rows.view(np.uint32))
    assert np.allclose(reference_rows[:4], [vectors[w] for w in sorted(vectors)], atol=0.02)
    reader = native.Reader(paths[True])
    assert reader.keys() == sorted(vectors) and reader.dim == 3
    assert reader.resolve_rows(words).tolist() == [0, 1, 2, 3, 0xFFFFFFFF]
    # full and trained storages in the same layout: one vtable for all FullNode tables
    for storage in ('full', 'trained'):
        files = {}
        for mimic in (False, True):
            _memb._writer_mimics_official_layout(mimic)
            try:
                builder = native.Builder(3, storage, 8)
                for word, vector in SIX_WORDS.items():
                    builder.add_word(word, np.array(vector, dtype=np.float32))
                files[mimic] = str(tmp_path / '{}_{}.bin'.format(storage, int(mimic)))
                builder.save(files[mimic])
            finally:
                _memb._writer_mimics_official_layout(False)
        batch = sorted(SIX_WORDS) + ['o']
        want = oracle.OracleReader(files[False]).batch_embedding(batch)
        assert np.array_equal(oracle.OracleReader(files[True]).batch_embedding(batch).view(np.uint32), want.view(np.uint32))
        assert native.Reader(files[True]).keys() == sorted(SIX_WORDS)
        if storage == 'full':
            assert os.path.getsize(files[True]) < os.path.getsize(files[False])


def test_device_builder_never_falls_back_to_the_host(native, tmp_path):
    # Builder(..., device=N) means the GPU does the bulk work; on a host without one it says so (the host
    # writer is what Builder() without a device is)
    if native.hip_device_count() > 0:
        pytest.skip('this host has a HIP device: tests/test_gpu_writer.py covers the device writer')
    from memb_amd import synthetic
    words = synthetic.make_words(300)
    vectors = synthetic.make_vectors(300, 8)
    builder = native.Builder(8, 'trained', 4, device=0)
    builder.add_words(words, vectors)
    with pytest.raises(RuntimeError, match='no HIP device available'):
        builder.save(str(tmp_path / 'never.bin'))
    # storages without device work ignore the argument
    for storage in ('uniform', 'full'):
        plain = native.Builder(8, storage, 8)
        on_device = native.Builder(8, storage, 8, device=0)
        for builder in (plain, on_device):
            builder.add_words(words, vectors)
        plain.save(str(tmp_path / 'a.bin'))
        on_device.save(str(tmp_path / 'b.bin'))
        assert open(str(tmp_path / 'a.bin'), 'rb').read() == open(str(tmp_path / 'b.bin'), 'rb').read()
