"""Parity of the HIP path (through the Reader -> C ABI) with the CPU checker and
the committed golden vectors. Bit-exact everywhere: trained decode is a table
lookup, uniform dequantisation is four correctly rounded IEEE fp32 operations
(north_star: bit-exact uniform, <= 1e-6 relative trained -- met with 0)."""
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, SIX_WORDS, bits_equal, golden_json

pytestmark = pytest.mark.gpu


def nan_aware_equal(a, b):
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    both_nan = np.isnan(a) & np.isnan(b)
    return a.shape == b.shape and bool(np.all(both_nan | (a.view(np.uint32) == b.view(np.uint32))))


def test_extension_is_loaded_and_device_present(native):
    assert native.hip_device_count() >= 1
    assert os.path.exists(native.HIP_LIBRARY_PATH)


@pytest.mark.parametrize('max_direct_bits', [0, 1, 3])
def test_golden_trained_models(native, max_direct_bits):
    # rows.npy come from the reference's own HuffmanTableDecoder (tests/golden/make_golden.py);
    # max_direct_bits = 1 is the reference's forced indirect-table test (src/tests.cpp:76-88)
    for entry in golden_json('models.json'):
        if entry['storage'] != 'trained':
            continue
        reader = native.Reader(os.path.join(GOLDEN, entry['file']), max_direct_decode_bits=max_direct_bits)
        rows = np.load(os.path.join(GOLDEN, entry['rows']))
        assert reader.keys() == entry['keys']
        assert bits_equal(reader.batch_embedding(entry['keys']), rows), entry['file']
        assert bits_equal(reader[entry['keys'][3]], rows[3])
        info = reader.info()
        assert info['storage'] == 3 and info['dim'] == entry['dim']
        if max_direct_bits:
            assert info['root_bits'] == min(max_direct_bits, info['max_code_bits'])


@pytest.mark.parametrize('storage', ['full', 'uniform', 'trained'])
def test_builder_round_trip_like_reference(native, storage):
    # reference src/tests.cpp:29-74
    path = os.path.join(GOLDEN, 'six_words_{}.bin'.format(storage))
    reader = native.Reader(path)
    checker = oracle.OracleReader(path)
    assert reader.keys() == sorted(SIX_WORDS)
    for word, vector in SIX_WORDS.items():
        embedding = reader[word]
        assert embedding.shape == (3,) and embedding.dtype == np.float32
        assert bits_equal(embedding, checker.word_embedding(word))
        np.testing.assert_allclose(embedding, vector, rtol=0.01, atol=1e-6)
    assert bits_equal(reader['o'], np.zeros(3, dtype=np.float32))
    batch = list(SIX_WORDS) + ['o', '', 'zzz']
    assert bits_equal(reader[batch], checker.batch_embedding(batch))


def test_batch_of_1025_equals_checker_serial_and_threaded(native):
    # reference src/tests.cpp:90-113
    path = os.path.join(GOLDEN, 'six_words_trained.bin')
    words = list(SIX_WORDS)
    batch = [words[i % len(words)] for i in range(1025)]
    expected = oracle.OracleReader(path, 1).batch_embedding(batch)
    assert bits_equal(native.Reader(path, 1)[batch], expected)
    assert bits_equal(native.Reader(path, 4)[batch], expected)


@pytest.mark.parametrize('bits,distribution', [(2, 'normal'), (4, 'normal'), (6, 'normal'), (8, 'normal'),
                                               (6, 'student'), (8, 'student')])
def test_trained_full_dump_and_random_batches(native, make_model, bits, distribution):
    path, words = make_model(20000, 300, 'trained', bits, distribution=distribution)
    reader = native.Reader(path)
    checker = oracle.OracleReader(path)
    keys = reader.keys()
    assert bits_equal(reader.batch_embedding(keys), checker.rows_embedding(np.arange(len(keys), dtype=np.uint32)))
    rng = np.random.default_rng(bits)
    batch = [keys[i] for i in rng.integers(0, len(keys), size=5000)]
    for position in rng.integers(0, len(batch), size=50):
        batch[position] = 'absent{}'.format(position)
    result = reader.batch_embedding(batch)
    assert bits_equal(result, checker.batch_embedding(batch))
    assert not result[[i for i, w in enumerate(batch) if w.startswith('absent')]].any()


@pytest.mark.parametrize('count', [0, 1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1000])
def test_ragged_batch_sizes(native, make_model, count):
    path, words = make_model(20000, 300, 'trained', 4)
    reader = native.Reader(path)
    checker = oracle.OracleReader(path)
    rows = np.random.default_rng(count).integers(0, len(words), size=count).astype(np.uint32)
    if count > 2:
        rows[count // 2] = 0xFFFFFFFF
    result = reader.rows_embedding(rows)
    assert result.shape == (count, 300)
    assert bits_equal(result, checker.rows_embedding(rows))
    assert reader.batch_embedding([]).shape == (0, 300)


@pytest.mark.parametrize('storage,bits', [('trained', 4), ('trained', 8), ('uniform', 8), ('full', 8)])
def test_small_batches_through_the_pinned_buffer(native, make_model, storage, bits):
    # batches of <= 512 words skip the staging copies (memb_hip_decode_rows): row
    # ids and rows share one pinned buffer and must not overlap, first call
    # included, into plain and strided outputs
    path, words = make_model(3000, 300, storage, bits)
    checker = oracle.OracleReader(path)
    for count in (512, 65, 1, 300, 512):
        reader = native.Reader(path)
        rows = np.random.default_rng(count).integers(0, len(words), size=count).astype(np.uint32)
        rows[count // 3] = 0xFFFFFFFF
        expected = checker.rows_embedding(rows)
        for _ in range(3):
            assert bits_equal(reader.rows_embedding(rows), expected), count
        keys = reader.keys()
        wide = np.full((count, 307), 7.0, dtype=np.float32)
        reader.batch_embedding_into([keys[r] if r < len(keys) else '?' for r in rows], wide, 5)
        assert bits_equal(wide[:, 5:305], expected), count
        assert (wide[:, :5] == 7.0).all() and (wide[:, 305:] == 7.0).all()
        # a reader on a device never decodes on the host unless asked to (host_below / device='cpu')
        assert reader.device == 0 and reader.host_rows_decoded == 0


def test_host_below_and_the_device_agree(native, make_model):
    # the opt-in host path for small host batches next to the HIP path of the same reader: same bits
    path, words = make_model(3000, 300, 'trained', 4)
    keys = sorted(words)
    on_device = native.Reader(path)
    mixed = native.Reader(path, host_below=64)
    for count in (1, 64, 65, 600):
        batch = keys[7:7 + count] + ['not-a-word']
        assert bits_equal(mixed[batch[:count]], on_device[batch[:count]])
    assert mixed.host_rows_decoded == 1 + 64 and on_device.host_rows_decoded == 0
    assert bits_equal(mixed[keys[5]], on_device[keys[5]])


def test_host_copy_ring_chunks_threads_and_slices(native, make_model, monkeypatch):
    # memb_hip_decode_rows streams results through a ring of pinned chunks that
    # host threads copy out; shrink chunks and slices so that a small batch runs
    # many of them, with and without copy threads, into dense and strided outputs
    path, words = make_model(20000, 300, 'trained', 4)
    checker = oracle.OracleReader(path)
    rows = np.random.default_rng(5).integers(0, len(words), size=9001).astype(np.uint32)
    rows[::50] = 0xFFFFFFFF
    expected = checker.rows_embedding(rows)
    for chunk_rows, threads, slice_words in ((0, 8, 0), (7, 3, 0), (64, 1, 0), (100, 8, 2500), (1, 2, 1000), (513, 0, 0), (9001, 64, 0)):
        if chunk_rows:
            monkeypatch.setenv('MEMB_HIP_COPY_CHUNK_ROWS', str(chunk_rows))
        else:
            monkeypatch.delenv('MEMB_HIP_COPY_CHUNK_ROWS', raising=False)
        monkeypatch.setenv('MEMB_HIP_COPY_THREADS', str(threads))
        if slice_words:
            monkeypatch.setenv('MEMB_HIP_SLICE_WORDS', str(slice_words))
        else:
            monkeypatch.delenv('MEMB_HIP_SLICE_WORDS', raising=False)
        reader = native.Reader(path)
        assert bits_equal(reader.rows_embedding(rows), expected), (chunk_rows, threads, slice_words)
        keys = reader.keys()
        wide = np.full((len(rows), 303), -1.0, dtype=np.float32)
        reader.batch_embedding_into([keys[r] if r < len(keys) else '?' for r in rows], wide, 2)
        assert bits_equal(wide[:, 2:302], expected), (chunk_rows, threads, slice_words)
        assert (wide[:, :2] == -1.0).all() and (wide[:, 302:] == -1.0).all()
        again = np.full((len(rows), 301), -2.0, dtype=np.float32)
        reader.rows_embedding_into(rows[100:], again[100:], 1)   # a row range of a larger matrix
        assert bits_equal(again[100:, 1:], expected[100:]) and (again[:100] == -2.0).all() and (again[:, 0] == -2.0).all()


def test_host_batches_cross_pcie_as_centroid_indices(native, make_model, monkeypatch):
    # trained storage + host buffers: the kernel writes rows of centroid indices, the host threads
    # that empty the pinned ring expand them (memb_hip.hip: decodeRowsAsKeys). Same bits as the
    # fp32 path (option host_expand = 0) and as the checker, for nibble and byte keys, odd
    # dimensions, tile geometries whose key tiles are not dword multiples, absent rows.
    for dim, bits, lanes in ((300, 4, 8), (300, 8, 8), (7, 2, 1), (301, 4, 5), (33, 6, 3), (150, 4, 64), (2, 4, 1)):
        path, words = make_model(2500, dim, 'trained', bits, seed=dim + bits)
        checker = oracle.OracleReader(path)
        rows = np.random.default_rng(dim).integers(0, len(words), size=1777).astype(np.uint32)
        rows[::13] = 0xFFFFFFFF
        rows[5] = len(words)
        expected = checker.rows_embedding(rows)
        monkeypatch.setenv('MEMB_HIP_LANES', str(lanes))
        for expand, chunk_rows in ((1, 0), (1, 5), (0, 0)):
            if chunk_rows:
                monkeypatch.setenv('MEMB_HIP_COPY_CHUNK_ROWS', str(chunk_rows))
            else:
                monkeypatch.delenv('MEMB_HIP_COPY_CHUNK_ROWS', raising=False)
            reader = native.Reader(path)
            reader.set_option('host_expand', expand)
            assert bits_equal(reader.rows_embedding(rows), expected), (dim, bits, lanes, expand, chunk_rows)
            wide = np.full((len(rows), dim + 5), 9.0, dtype=np.float32)
            keys = reader.keys()
            reader.batch_embedding_into([keys[r] if r < len(keys) else '?' for r in rows], wide, 3)
            assert bits_equal(wide[:, 3:3 + dim], expected), (dim, bits, lanes, expand, chunk_rows)
            assert (wide[:, :3] == 9.0).all() and (wide[:, 3 + dim:] == 9.0).all()


def test_tokenizer_embedding_host_and_device(native, make_model):
    # reference python/memb/reader.py:89-111: row idx of the matrix belongs to word_index[word] == idx,
    # row 0 and unknown words are zeros, num_words cuts the table (idx < num_words)
    import torch
    path, words = make_model(3000, 300, 'trained', 4)
    path2, words2 = make_model(2000, 300, 'trained', 6, seed=5)
    reader, other = native.Reader(path), native.Reader(path2)
    checker = oracle.OracleReader(path)

    class Tokenizer:
        word_index = {words[10]: 1, 'not in the model': 2, words[7]: 3, words[2999]: 5}
        num_words = None

    for num_words, expected_words in ((None, ['', words[10], 'not in the model', words[7], '', words[2999]]),
                                      (4, ['', words[10], 'not in the model', words[7]])):
        Tokenizer.num_words = num_words
        expected = checker.batch_embedding(expected_words)
        host = reader.tokenizer_embedding(Tokenizer)
        assert bits_equal(host, expected) and not host[0].any() and not host[2].any()
        device = reader.tokenizer_embedding_device(Tokenizer)
        assert device.is_cuda and bits_equal(device.cpu().numpy(), expected)
        layer = torch.nn.Embedding.from_pretrained(device)   # takes the device tensor as is
        assert layer.weight.data_ptr() == device.data_ptr()
        union = native.ReadersUnion([reader, other], 'concatenate')
        merged = union.tokenizer_embedding_device(Tokenizer)
        assert bits_equal(merged.cpu().numpy(), union.tokenizer_embedding(Tokenizer))
        assert bits_equal(merged[:, :300].cpu().numpy(), expected)


def test_device_entry_point_can_be_captured_into_a_graph(native, make_model):
    # memb_hip_decode_rows_device only enqueues (no allocation, no synchronisation once a kernel
    # variant has been launched once), so a HIP graph can hold it; replays give the same bits
    import torch
    path, words = make_model(20000, 300, 'trained', 4)
    reader = native.Reader(path)
    checker = oracle.OracleReader(path)
    host_rows = np.random.default_rng(8).integers(0, len(words), size=5000).astype(np.uint32)
    host_rows[::9] = 0xFFFFFFFF
    rows = torch.from_numpy(host_rows.view(np.int32)).cuda()
    out = torch.empty((len(host_rows), 300), dtype=torch.float32, device='cuda')
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        reader.rows_embedding_device(rows, out=out)   # first launch configures the kernel
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        reader.rows_embedding_device(rows, out=out)
    expected = checker.rows_embedding(host_rows)
    for _ in range(3):
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert bits_equal(out.cpu().numpy(), expected)


def test_arbitrary_prefix_codes_through_the_c_abi(native):
    # Storages the reference's writer cannot produce (its k-means prunes rare clusters, which bounds
    # the code lengths): random complete prefix codes with up to 255 symbols and code lengths up to
    # 16 bits -- the format's maximum, PrefixCode::code is a uint16 (reference src/prefix_code.h:10-13)
    # -- handed to the C ABI as a crafted description, for several first-level table widths and
    # batch sizes on both sides of the small-batch threshold. Expected rows: the symbols themselves.
    import ctypes
    from test_oracle_vs_reference import random_code
    library = ctypes.CDLL(native.HIP_LIBRARY_PATH)
    library.memb_hip_last_error.restype = ctypes.c_char_p

    class Desc(ctypes.Structure):
        _fields_ = [('dim', ctypes.c_uint32), ('n_rows', ctypes.c_uint64),
                    ('packed_values', ctypes.c_void_p), ('packed_values_bytes', ctypes.c_uint64),
                    ('value_offsets', ctypes.c_void_p),
                    ('keys', ctypes.c_void_p), ('n_keys', ctypes.c_uint32),
                    ('size_offsets', ctypes.c_void_p), ('n_size_offsets', ctypes.c_uint32),
                    ('centroids', ctypes.c_void_p), ('n_centroids', ctypes.c_uint32),
                    ('max_direct_bits', ctypes.c_uint32)]

    rng = np.random.default_rng(2024)
    codec = oracle.Codec('oracle')
    longest = 0
    for trial, (symbols, dim, n_rows) in enumerate(((255, 61, 700), (200, 300, 40), (17, 5, 900), (2, 33, 64), (90, 128, 600), (254, 8, 513))):
        while True:
            keys, lengths, size_offsets = random_code(rng, symbols)
            if trial != 0 or max(lengths) >= 15:   # the first storage must reach (nearly) the longest codes
                break
        longest = max(longest, max(lengths))
        keys = np.ascontiguousarray(keys, dtype=np.uint8)
        size_offsets = np.ascontiguousarray(size_offsets, dtype=np.uint32)
        codes, bits = codec.canonical_codes(keys, lengths)
        centroids = rng.standard_normal(255).astype(np.float32)
        # symbol draws that favour the long codes (uniform over symbols, not over probability mass)
        picks = rng.integers(0, symbols, size=(n_rows, dim))
        symbol_rows = keys[picks]
        streams, offsets = [], []
        position = 0
        for row in symbol_rows:
            stream = codec.bitstream_pack(codes[row], bits[row])
            offsets.append(position)
            streams.append(stream)
            position += len(stream)
        packed = np.concatenate(streams) if position else np.zeros(1, dtype=np.uint8)
        offsets = np.array(offsets, dtype=np.uint32)
        expected = centroids[symbol_rows]
        for max_direct_bits in (0, 1, 5, 12):
            desc = Desc(dim, n_rows, packed.ctypes.data, position, offsets.ctypes.data, keys.ctypes.data, len(keys),
                        size_offsets.ctypes.data, len(size_offsets), centroids.ctypes.data, len(centroids), max_direct_bits)
            context = ctypes.c_void_p()
            assert library.memb_hip_ctx_create_trained(ctypes.byref(context), 0, ctypes.byref(desc)) == 0, \
                library.memb_hip_last_error()
            for count in (n_rows, min(n_rows, 100)):
                ids = rng.permutation(n_rows)[:count].astype(np.uint32)
                ids[::7] = 0xFFFFFFFF
                out = np.full((count, dim), 5.0, dtype=np.float32)
                code = library.memb_hip_decode_rows(
                    context, ids.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(count),
                    out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(dim), ctypes.c_size_t(0))
                assert code == 0, library.memb_hip_last_error()
                want = np.where((ids == 0xFFFFFFFF)[:, None], np.float32(0), expected[np.minimum(ids, n_rows - 1)])
                assert bits_equal(out, want), (trial, symbols, dim, max_direct_bits, count, max(lengths))
            library.memb_hip_ctx_destroy(context)
    assert longest >= 15


@pytest.mark.parametrize('dim,bits,count', [(4096, 8, 120), (9000, 8, 40), (20000, 4, 30), (60000, 8, 10), (100000, 2, 6)])
def test_very_wide_rows(native, tmp_path, dim, bits, count):
    # the reference reads any dimension; here a row's bitstream has to fit into LDS (~140 KB). Wide
    # rows take more lanes per word (up to one word per wavefront), rows longer than 65535 bits a
    # 32-bit segment index, and the index-building pass fewer words per wavefront
    from memb_amd import synthetic
    path = str(tmp_path / 'wide.bin')
    words = synthetic.build_file(path, count, dim, 'trained', bits, seed=dim)
    reader = native.Reader(path)
    checker = oracle.OracleReader(path)
    info = reader.info()
    assert info['lanes_per_word'] * info['segment_symbols'] >= dim
    batch = sorted(words)[::-1] + ['not there'] + sorted(words)[:3]
    assert bits_equal(reader.batch_embedding(batch), checker.batch_embedding(batch))
    assert bits_equal(reader[batch[0]], checker.word_embedding(batch[0]))
    many = (batch * (600 // len(batch) + 1))[:600]   # past the small-batch path: centroid indices over PCIe
    assert bits_equal(reader.batch_embedding(many), checker.batch_embedding(many))


def test_degenerate_models(native, tmp_path):
    # one-symbol codes (every stream is empty), two-symbol codes, a single word, heavy tails,
    # constant / subnormal / near-overflow uniform rows, special values in full storage, odd words
    rng = np.random.default_rng(0)
    words = ['w%04d' % i for i in range(300)]
    special = rng.standard_normal((300, 40)).astype(np.float32)
    special[0, 0], special[1, 1], special[2, 2], special[3, 3] = np.inf, -np.inf, np.nan, -0.0
    cases = []
    for bits in (1, 4, 8):
        cases += [
            ('trained', bits, words, np.full((300, 40), 0.25, dtype=np.float32)),
            ('trained', bits, words, rng.choice(np.array([-1.0, 2.0], dtype=np.float32), size=(300, 40))),
            ('trained', bits, words[:1], rng.standard_normal((1, 40)).astype(np.float32)),
            ('trained', bits, words, rng.standard_t(1.5, size=(300, 64)).astype(np.float32)),
        ]
    cases += [
        ('uniform', 8, words, np.full((300, 40), 3.0, dtype=np.float32)),
        ('uniform', 8, words, (rng.standard_normal((300, 40)) * 1e-40).astype(np.float32)),
        ('uniform', 8, words, (rng.standard_normal((300, 40)) * 1e37).astype(np.float32)),
        ('full', 8, words, special),
        ('trained', 4, ['\u00e9t\u00e9', '\u65e5\u672c', 'a b', '', '\U0001F600', 'z' * 300], rng.standard_normal((6, 8)).astype(np.float32)),
    ]
    for index, (storage, bits, names, vectors) in enumerate(cases):
        path = str(tmp_path / 'degenerate_{}.bin'.format(index))
        builder = native.Builder(vectors.shape[1], storage, bits)
        builder.add_words(names, vectors)
        builder.save(path)
        reader, checker = native.Reader(path), oracle.OracleReader(path)
        batch = list(names) + ['nope', '\u00e9', '\u65e5']
        assert nan_aware_equal(reader.batch_embedding(batch), checker.batch_embedding(batch)), (index, storage, bits)
        long_batch = (batch * (700 // len(batch) + 1))[:700]
        assert nan_aware_equal(reader.batch_embedding(long_batch), checker.batch_embedding(long_batch)), (index, storage, bits)


def test_omitted_default_scalars_on_device(native, tmp_path):
    # see tests/test_host_logic.py::test_files_with_omitted_default_scalars
    from memb_amd import _memb
    rng = np.random.default_rng(6)
    words = ['w%03d' % i for i in range(200)]
    vectors = rng.standard_normal((200, 24)).astype(np.float32)
    vectors[::3] = np.abs(vectors[::3])
    vectors[::3, 0] = 0.0            # min exactly 0.0
    vectors[1::3] = -np.abs(vectors[1::3])
    vectors[1::3, 5] = 0.0           # max exactly 0.0
    vectors[10] = 0.0
    _memb._writer_mimics_official_layout(True)
    try:
        builder = native.Builder(24, 'uniform', 8)
        builder.add_words(words, vectors)
        path = str(tmp_path / 'omitted.bin')
        builder.save(path)
    finally:
        _memb._writer_mimics_official_layout(False)
    batch = words[::-1] + ['nope']
    assert bits_equal(native.Reader(path).batch_embedding(batch), oracle.OracleReader(path).batch_embedding(batch))


def test_union_concatenation_in_one_launch(native, make_model, monkeypatch):
    # memb_hip_decode_rows_union_device: two trained models of one geometry decoded by one kernel that
    # writes the merged rows whole. Same bits as one launch per reader (option union_fused = 0) and as
    # numpy's concatenation of the checker's rows; nibble and byte keys, two-level tables, batches
    # that end inside a tile, words missing from one or both models, a model paired with itself.
    import torch
    cases = [((20000, 300, 4, 1234), (15000, 300, 4, 99)),      # nibble keys
             ((20000, 300, 6, 1234), (15000, 300, 8, 99)),      # byte keys, the second with two-level tables
             ((3000, 64, 2, 5), (3000, 64, 4, 6)),              # small dim, different codebooks
             ((3000, 100, 4, 5), (2500, 100, 4, 6)),            # 25 pieces per half row
             ((1500, 1024, 4, 5), (1500, 1024, 6, 6))]          # wide rows, mixed key formats
    for first, second in cases:
        path_a, words_a = make_model(first[0], first[1], 'trained', first[2], seed=first[3])
        path_b, words_b = make_model(second[0], second[1], 'trained', second[2], seed=second[3])
        dim = first[1]
        readers = [native.Reader(path_a), native.Reader(path_b)]
        checkers = [oracle.OracleReader(path_a), oracle.OracleReader(path_b)]
        union = native.ReadersUnion(readers, 'concatenate')
        rng = np.random.default_rng(first[2])
        for count in (1, 7, 8, 9, 513, 4001):
            pool = sorted(set(words_a[:3000]) | set(words_b[:3000]))
            batch = [pool[i] for i in rng.integers(0, len(pool), size=count)]
            batch[::11] = ['in neither'] * len(batch[::11])
            expected = np.concatenate([checker.batch_embedding(batch) for checker in checkers], axis=1)
            readers[0].set_option('union_fused', 1)
            fused = union.batch_embedding_device(batch).cpu().numpy()
            readers[0].set_option('union_fused', 0)
            separate = union.batch_embedding_device(batch).cpu().numpy()
            assert fused.shape == (count, 2 * dim)
            assert bits_equal(fused, expected), (first, second, count)
            assert bits_equal(separate, expected), (first, second, count)
            # the 'average' mode through the same kernel: numpy.mean of the checker's rows, bit for bit
            mean = native.ReadersUnion(readers, 'average')
            expected_mean = np.mean([checker.batch_embedding(batch) for checker in checkers], axis=0)
            readers[0].set_option('union_fused', 1)
            assert bits_equal(mean.batch_embedding_device(batch).cpu().numpy(), expected_mean), (first, second, count)
            readers[0].set_option('union_fused', 0)
            assert bits_equal(mean.batch_embedding_device(batch).cpu().numpy(), expected_mean), (first, second, count)
        readers[0].set_option('union_fused', 1)
        twice = native.ReadersUnion([readers[0], readers[0]], 'concatenate').batch_embedding_device(words_a[:100])
        assert bits_equal(twice[:, :dim].cpu().numpy(), twice[:, dim:].cpu().numpy())
    # three and four readers through the same kernel (and five: no kernel, one launch per reader)
    models = [make_model(4000, 300, 'trained', bits, seed=40 + bits) for bits in (2, 4, 4, 2, 4)]
    models[2] = make_model(5000, 300, 'trained', 4, seed=77)
    many_readers = [native.Reader(path) for path, _ in models]
    many_checkers = [oracle.OracleReader(path) for path, _ in models]
    pool = sorted(set().union(*[set(words[:1500]) for _, words in models]))
    batch = [pool[i] for i in np.random.default_rng(9).integers(0, len(pool), size=2500)] + ['nowhere']
    for count in (3, 4, 5):
        rows = [checker.batch_embedding(batch) for checker in many_checkers[:count]]
        concatenated = native.ReadersUnion(many_readers[:count], 'concatenate').batch_embedding_device(batch)
        assert bits_equal(concatenated.cpu().numpy(), np.concatenate(rows, axis=1)), count
        averaged = native.ReadersUnion(many_readers[:count], 'average').batch_embedding_device(batch)
        assert bits_equal(averaged.cpu().numpy(), np.mean(rows, axis=0)), count
    # the C entry point says so when it has no kernel for a combination; the Python layer then launches per reader
    from memb_amd import _memb
    path_c, words_c = make_model(2000, 100, 'trained', 4, seed=3)
    path_u, words_u = make_model(2000, 300, 'uniform', 8, seed=3)
    odd = [native.Reader(path_c), native.Reader(path_u)]
    rows = [torch.from_numpy(reader.resolve_rows(words_c[:50]).view(np.int32)).cuda() for reader in odd]
    out = torch.empty((50, 400), dtype=torch.float32, device='cuda')
    assert _memb.union_rows_to_device([reader._impl for reader in odd], [r.data_ptr() for r in rows], [0, 100], 50,
                                      out.data_ptr(), 400, 0, False) is False
    mixed = native.ReadersUnion(odd, 'concatenate')
    assert bits_equal(mixed.batch_embedding_device(words_c[:50]).cpu().numpy(), mixed.batch_embedding(words_c[:50]))


@pytest.mark.parametrize('dim', [1, 3, 4, 5, 8, 64, 100, 302, 1024])
def test_other_dimensions(native, make_model, dim):
    for storage, bits in (('trained', 4), ('trained', 8), ('uniform', 8), ('full', 8)):
        path, words = make_model(700, dim, storage, bits, seed=dim)
        reader = native.Reader(path)
        checker = oracle.OracleReader(path)
        batch = sorted(words)[::-1] + ['nope']
        assert nan_aware_equal(reader.batch_embedding(batch), checker.batch_embedding(batch)), (storage, bits)


@pytest.mark.parametrize('bits', [1, 4, 8])
def test_uniform_is_bit_exact(native, make_model, bits):
    # config 1 of BASELINE.json: 1k-word vocabulary, dim 300, uniform; all keys + 10 % misses
    path, words = make_model(1000, 300, 'uniform', bits)
    reader = native.Reader(path)
    checker = oracle.OracleReader(path)
    batch = list(reader.keys()) + ['missing{}'.format(i) for i in range(100)]
    np.random.default_rng(1).shuffle(batch)
    result = reader.batch_embedding(batch)
    assert bits_equal(result, checker.batch_embedding(batch))


def test_uniform_expression_vectors_on_device(native, tmp_path):
    # every (min, max, value, levels) of tests/golden/uniform_expr.json through the kernel:
    # one hand-made uniform file per `levels`, rows = the (min, max) pairs
    from memb_amd import _memb
    cases = golden_json('uniform_expr.json')
    # (levels 0 = a file without the quantization_levels field: division by zero, +-inf and NaN rows;
    #  the vectors also hold max < min and all-subnormal rows)
    for levels in (0, 1, 2, 16, 255):
        subset = [c for c in cases if c[3] == levels]
        pairs = sorted({(c[0], c[1]) for c in subset})
        values = sorted({c[2] for c in subset})
        # Builder quantises real vectors, so drive the C ABI directly with a crafted description
        import ctypes
        library = ctypes.CDLL(native.HIP_LIBRARY_PATH)
        library.memb_hip_last_error.restype = ctypes.c_char_p

        class Row(ctypes.Structure):
            _fields_ = [('values', ctypes.c_void_p), ('n_values', ctypes.c_uint32),
                        ('min_value', ctypes.c_float), ('max_value', ctypes.c_float)]

        class Desc(ctypes.Structure):
            _fields_ = [('dim', ctypes.c_uint32), ('n_rows', ctypes.c_uint64), ('rows', ctypes.c_void_p),
                        ('quantization_levels', ctypes.c_uint8)]

        values = values + [values[-1]] * (-len(values) % 4)   # whole 16-byte output pieces: the vector kernels
        payload = np.array(values, dtype=np.uint8)
        rows = (Row * len(pairs))()
        for i, (low, high) in enumerate(pairs):
            rows[i] = Row(payload.ctypes.data, len(values), np.uint32(low).view(np.float32), np.uint32(high).view(np.float32))
        desc = Desc(len(values), len(pairs), ctypes.addressof(rows), levels)
        context = ctypes.c_void_p()
        assert library.memb_hip_ctx_create_uniform(ctypes.byref(context), 0, ctypes.byref(desc)) == 0, library.memb_hip_last_error()
        ids = np.arange(len(pairs), dtype=np.uint32)
        out = np.empty((len(pairs), len(values)), dtype=np.float32)
        code = library.memb_hip_decode_rows(
            context, ids.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(len(pairs)),
            out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(len(values)), ctypes.c_size_t(0))
        assert code == 0, library.memb_hip_last_error()
        # the same rows many times over, with missing rows: a large batch (many tiles per CU, a ragged last tile)
        many = np.tile(ids, 70000 // len(ids) + 1)
        many[5::1001] = 0xFFFFFFFF
        out_many = np.empty((len(many), len(values)), dtype=np.float32)
        code = library.memb_hip_decode_rows(
            context, many.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(len(many)),
            out_many.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(len(values)), ctypes.c_size_t(0))
        assert code == 0, library.memb_hip_last_error()
        present = many != 0xFFFFFFFF
        assert nan_aware_equal(out_many[present], out[many[present]]), levels
        assert not out_many[~present].view(np.uint32).any()
        library.memb_hip_ctx_destroy(context)
        expected = {(c[0], c[1], c[2]): c[4] for c in subset}
        for i, (low, high) in enumerate(pairs):
            for j, value in enumerate(values):
                want = np.uint32(expected[(low, high, value)]).view(np.float32)
                got = out[i, j]
                assert (np.isnan(want) and np.isnan(got)) or got.view(np.uint32) == np.uint32(expected[(low, high, value)]), \
                    (low, high, value, levels)


def test_uniform_division_for_every_level_count(native):
    """Every divisor 1 .. 255, every weight 0 .. 255, rows whose (min, max) make range * weight ordinary, tiny, huge,
    subnormal, negative and zero -- both uniform kernels against the reference's expression as compiled with its own
    flags (oracle/uniform_expr.cpp), bit for bit. (Written for round 4's reciprocal + FMA division, which was exact and
    not faster and is gone again: hip_rowwise_kernels.h; the vectors in tests/golden cover five level counts.)"""
    import ctypes
    library = ctypes.CDLL(native.HIP_LIBRARY_PATH)
    library.memb_hip_last_error.restype = ctypes.c_char_p

    class Row(ctypes.Structure):
        _fields_ = [('values', ctypes.c_void_p), ('n_values', ctypes.c_uint32),
                    ('min_value', ctypes.c_float), ('max_value', ctypes.c_float)]

    class Desc(ctypes.Structure):
        _fields_ = [('dim', ctypes.c_uint32), ('n_rows', ctypes.c_uint64), ('rows', ctypes.c_void_p),
                    ('quantization_levels', ctypes.c_uint8)]

    rng = np.random.default_rng(255)
    payload = np.arange(256, dtype=np.uint8)
    lows = (rng.standard_normal(160) * 0.5).astype(np.float32)
    highs = (lows + np.abs(rng.standard_normal(160)).astype(np.float32) * np.float32(1.5)).astype(np.float32)
    # ranges whose products with the weights straddle the fast path's limits (2^-100, 2^100), tiny, huge, negative, zero
    for low, high in ((0.0, 2.0 ** -107), (0.0, 2.0 ** -100), (-2.0 ** -104, 2.0 ** -104), (1.0, 1.0), (3.0, -2.0),
                      (0.0, 2.0 ** 93), (-2.0 ** 99, 2.0 ** 99), (0.0, 2.0 ** 120), (1e-30, 1e-29), (-1e30, 1e30),
                      (0.0, 1e-44), (2.0 ** -126, 2.0 ** -125), (0.25, 0.25 + 2.0 ** -20), (-0.0, 0.0)):
        lows = np.append(lows, np.float32(low))
        highs = np.append(highs, np.float32(high))
    count = len(lows)
    ids = np.tile(np.arange(count, dtype=np.uint32), 400)   # 69 600 rows: the persistent kernel; the first `count` also alone
    for levels in range(1, 256):
        rows = (Row * count)()
        for i in range(count):
            rows[i] = Row(payload.ctypes.data, 256, float(lows[i]), float(highs[i]))
        desc = Desc(256, count, ctypes.addressof(rows), levels)
        context = ctypes.c_void_p()
        assert library.memb_hip_ctx_create_uniform(ctypes.byref(context), 0, ctypes.byref(desc)) == 0, library.memb_hip_last_error()
        try:
            expected = np.stack([oracle.uniform_expression(lows[i], highs[i], levels, payload) for i in range(count)])
            for batch in (ids[:count], ids):
                out = np.empty((len(batch), 256), dtype=np.float32)
                assert library.memb_hip_decode_rows(
                    context, batch.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(len(batch)),
                    out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(256), ctypes.c_size_t(0)) == 0, library.memb_hip_last_error()
                assert nan_aware_equal(out, expected[batch]), levels
        finally:
            library.memb_hip_ctx_destroy(context)


def test_strided_output_leaves_other_columns_alone(native, make_model):
    path, words = make_model(20000, 300, 'trained', 4)
    reader = native.Reader(path)
    checker = oracle.OracleReader(path)
    batch = sorted(words)[100:1500] + ['zz-missing']
    for width, col_off in ((600, 300), (604, 4), (301, 1), (900, 300)):
        out = np.full((len(batch), width), 7.5, dtype=np.float32)
        reader.batch_embedding_into(batch, out, col_off)
        assert bits_equal(out[:, col_off:col_off + 300], checker.batch_embedding(batch))
        untouched = np.delete(out, np.s_[col_off:col_off + 300], axis=1)
        assert (untouched == 7.5).all()


def test_row_strided_views_are_filled_in_place(native, make_model):
    # column ranges and row ranges of a wider matrix are written where they are (no temporary copy)
    path, words = make_model(20000, 300, 'trained', 4)
    reader = native.Reader(path)
    checker = oracle.OracleReader(path)
    batch = sorted(words)[:900] + ['zz-missing']
    want = checker.batch_embedding(batch)
    merged = np.full((len(batch), 640), -3.0, dtype=np.float32)
    reader.batch_embedding_into(batch, merged[:, 320:620], 0)      # view: ld 640, 300 columns
    assert bits_equal(merged[:, 320:620], want)
    assert (merged[:, :320] == -3.0).all() and (merged[:, 620:] == -3.0).all()
    rows = reader.resolve_rows(batch)
    tall = np.full((len(batch) + 10, 300), -3.0, dtype=np.float32)
    reader.rows_embedding_into(rows, tall[5:-5], 0)                  # row range
    assert bits_equal(tall[5:-5], want) and (tall[:5] == -3.0).all() and (tall[-5:] == -3.0).all()
    reader.rows_embedding_into(rows[:1], merged[:1, 10:310], 0)      # a single row of a view
    assert bits_equal(merged[:1, 10:310], want[:1])


def test_readers_union_concatenate_and_average(native, make_model):
    # config 5 of BASELINE.json in small: two models with overlapping keys, (n, 600) output
    path_a, words_a = make_model(20000, 300, 'trained', 4)
    path_b, words_b = make_model(9000, 300, 'trained', 4, seed=77)
    readers = [native.Reader(path_a), native.Reader(path_b)]
    checkers = [oracle.OracleReader(path_a), oracle.OracleReader(path_b)]
    batch = sorted(words_a)[:3000:2] + ['only-nowhere'] + sorted(words_b)[:500]
    expected = [c.batch_embedding(batch) for c in checkers]
    concat = native.ReadersUnion(readers, 'concatenate')
    assert concat.dim == 600
    result = concat[batch]
    assert result.shape == (len(batch), 600)
    assert bits_equal(result, np.concatenate(expected, axis=-1))  # reference python/memb/readers_union.py:32
    assert bits_equal(concat[batch[5]], np.concatenate([e[5] for e in expected]))
    average = native.ReadersUnion(readers, 'average')
    assert bits_equal(average[batch], np.mean(expected, axis=0))  # reference python/memb/readers_union.py:18
    assert concat.keys() == sorted(set(words_a) | set(words_b))


def test_union_merged_on_the_device(native, make_model):
    # device-side concatenate / average must equal numpy's merge of the CPU checker's rows, bit for bit
    import torch
    specs = [(20000, 'trained', 4, 1234), (9000, 'trained', 4, 77), (700, 'uniform', 8, 300), (20000, 'trained', 8, 1234)]
    models = [make_model(count, 300, storage, bits, seed=seed) for count, storage, bits, seed in specs]
    readers = [native.Reader(path) for path, _ in models]
    checkers = [oracle.OracleReader(path) for path, _ in models]
    vocabulary = sorted(set(models[0][1]) | set(models[1][1]) | set(models[2][1]))
    rng = np.random.default_rng(21)
    batch = [vocabulary[i] for i in rng.integers(0, len(vocabulary), size=3001)] + ['nowhere', '']
    expected = [checker.batch_embedding(batch) for checker in checkers]
    for count in (2, 3, 4):
        union = native.ReadersUnion(readers[:count], 'average')
        merged = union.batch_embedding_device(batch)
        assert merged.device.type == 'cuda' and merged.shape == (len(batch), 300)
        assert bits_equal(merged.cpu().numpy(), np.mean(expected[:count], axis=0)), count
        assert bits_equal(merged.cpu().numpy(), union[batch])   # the reference's host-side merge
        concat = native.ReadersUnion(readers[:count], 'concatenate').batch_embedding_device(batch)
        assert bits_equal(concat.cpu().numpy(), np.concatenate(expected[:count], axis=-1)), count
    # hand-off without a copy
    exported = torch.from_dlpack(merged)
    assert exported.data_ptr() == merged.data_ptr()
    single = readers[0].batch_embedding_device(batch)
    assert bits_equal(single.cpu().numpy(), expected[0])


def test_epilogue_scalar_paths(native, make_model):
    # odd dimensions take the scalar output path; accumulate + divide there too
    import torch
    for dim, storage, bits in ((5, 'trained', 4), (5, 'trained', 8), (7, 'uniform', 8), (3, 'full', 8)):
        path_a, words = make_model(700, dim, storage, bits, seed=1)
        path_b, _ = make_model(700, dim, storage, bits, seed=2)
        readers = [native.Reader(path_a), native.Reader(path_b)]
        checkers = [oracle.OracleReader(path_a), oracle.OracleReader(path_b)]
        batch = sorted(words)[::3] + ['missing']
        expected = np.mean([c.batch_embedding(batch) for c in checkers], axis=0)
        merged = native.ReadersUnion(readers, 'average').batch_embedding_device(batch)
        assert bits_equal(merged.cpu().numpy(), expected), (dim, storage, bits)


def test_device_resident_lookup_matches_host_path(native, make_model):
    import torch
    path, words = make_model(20000, 300, 'trained', 6)
    reader = native.Reader(path)
    checker = oracle.OracleReader(path)
    rows = np.random.default_rng(8).integers(0, len(words), size=4097).astype(np.uint32)
    rows[17] = 0xFFFFFFFF
    device_rows = torch.from_numpy(rows.view(np.int32)).cuda()
    out = reader.rows_embedding_device(device_rows)
    torch.cuda.synchronize()
    assert bits_equal(out.cpu().numpy(), checker.rows_embedding(rows))
    wide = torch.full((len(rows), 640), -1.0, device='cuda')
    reader.rows_embedding_device(device_rows, out=wide, col_off=40)
    torch.cuda.synchronize()
    assert bits_equal(wide[:, 40:340].cpu().numpy(), checker.rows_embedding(rows))
    assert bool((wide[:, :40] == -1).all()) and bool((wide[:, 340:] == -1).all())


def test_large_tables_take_two_tiles_per_wavefront_by_default(native, make_model):
    """The one rule of memb_hip.hip's oneTileSteps that no BASELINE model reaches: a model whose table and codebook are 16 KiB
    and more (8-bit: a 13-bit first level) decodes two tiles per wavefront behind one copy once the batch has two tiles
    per resident wavefront slot -- DEFAULT options, against the checker."""
    import torch
    path, words = make_model(150000, 300, 'trained', 8)   # (N(0, 0.4^2): codes up to 13 bits, a 32 KiB first level)
    reader = native.Reader(path, device=0)
    checker = oracle.OracleReader(path, os.cpu_count() or 1)
    slots = 32 * torch.cuda.get_device_properties(0).multi_processor_count
    tile = 64 // reader.info()['lanes_per_word']
    for count in (2 * slots * tile + 5, 2 * slots * tile - 8):
        facts = reader.info(count)
        assert facts['kernel'].startswith('decode_trained<') and facts['table_entries'] * 4 >= 16 * 1024
        assert facts['tiles_per_wavefront'] == (2 if count >= 2 * slots * tile else 1), (count, facts['tiles_per_wavefront'])
        rng = np.random.default_rng(count)
        rows = rng.integers(0, len(words), size=count).astype(np.uint32)
        rows[rng.integers(0, count, size=count // 100)] = 0xFFFFFFFF
        rows[:100000] = np.arange(100000, dtype=np.uint32)
        ids = torch.from_numpy(rows.view(np.int32)).cuda()
        assert bits_equal(reader.rows_embedding_device(ids).cpu().numpy(), checker.rows_embedding(rows)), count


def test_batch_split_like_two_ranks(native, make_model):
    # the N > 1 split (memb_amd/sharding.py) through the HIP path, slices concatenated on the host
    from memb_amd.sharding import lookup_shard
    path, words = make_model(20000, 300, 'trained', 2)
    reader = native.Reader(path)
    batch = sorted(words)[:5001]
    pieces = [lookup_shard(reader, batch, rank, 2) for rank in range(2)]
    assert bits_equal(np.concatenate(pieces), reader.batch_embedding(batch))


def test_boundary_argument_errors(native, make_model):
    import ctypes
    import torch
    path, words = make_model(700, 8, 'trained', 4, seed=8)
    reader = native.Reader(path)
    rows = torch.zeros(4, dtype=torch.int32, device='cuda')
    with pytest.raises(RuntimeError, match='ld must be at least col_off'):
        reader.rows_embedding_device(rows, out=torch.empty((4, 8), device='cuda'), col_off=4)
    with pytest.raises(TypeError):
        reader.rows_embedding_device(rows.cpu())
    if torch.cuda.device_count() > 1:   # tensors on another GPU than the reader's are refused, not decoded into
        with pytest.raises(ValueError, match='must be on cuda:0'):
            reader.rows_embedding_device(rows.to('cuda:1'))
    before = torch.cuda.current_device()
    with pytest.raises(TypeError):
        reader.rows_embedding_device(rows.to(torch.int64))
    with pytest.raises(TypeError):
        reader.rows_embedding_device(rows, out=torch.empty((4, 8), dtype=torch.float64, device='cuda'))
    with pytest.raises(RuntimeError):
        reader.batch_embedding_into(['a'], np.zeros((1, 4), dtype=np.float32), 0)   # narrower than dim
    # the measurement switches (skip the decode, skip the output, unused LDS ...) exist in builds with -DMEMB_HIP_MEASURE only
    for name in ('debug', 'lds_pad', 'no_such_option'):
        with pytest.raises(RuntimeError, match='unknown option'):
            reader.set_option(name, 1)
    with pytest.raises(RuntimeError, match='out of range'):
        reader.set_option('fine_lanes', 3)
    library = ctypes.CDLL(native.HIP_LIBRARY_PATH)
    library.memb_hip_last_error.restype = ctypes.c_char_p
    context = ctypes.c_void_p(reader._impl.context_handle())
    out = torch.empty((4, 8), device='cuda')
    call = library.memb_hip_decode_rows_device_ex
    args = [context, ctypes.c_void_p(rows.data_ptr()), ctypes.c_size_t(4), ctypes.c_void_p(out.data_ptr()),
            ctypes.c_size_t(8), ctypes.c_size_t(0), ctypes.c_void_p(0)]
    assert call(*args, ctypes.c_uint32(4), ctypes.c_float(0.0)) == 1 and b'flags' in library.memb_hip_last_error()   # 1 and 2 exist
    assert call(*args, ctypes.c_uint32(2), ctypes.c_float(0.0)) == 0   # MEMB_HIP_ROWS_IN_RANDOM_ORDER: a hint, same rows
    assert call(*args, ctypes.c_uint32(0), ctypes.c_float(float('nan'))) == 1
    assert call(*args, ctypes.c_uint32(0), ctypes.c_float(0.0)) == 0
    torch.cuda.synchronize()
    assert torch.cuda.current_device() == before   # entry points leave the caller's current device alone
    assert bits_equal(out.cpu().numpy(), oracle.OracleReader(path).rows_embedding(np.zeros(4, dtype=np.uint32)))
    total = ctypes.c_uint64(0)
    ids = np.array([0, 0xFFFFFFFF, 5], dtype=np.uint32)
    assert library.memb_hip_algorithmic_bytes(context, ids.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(3), ctypes.byref(total)) == 0
    checker = oracle.OracleReader(path)
    assert total.value == 3 * (4 + 32) + 2 * 4 + checker.stream_bytes(0) + checker.stream_bytes(5)   # SURVEY 8d formula


def test_empty_models(native, tmp_path):
    # no words at all: every lookup is a miss (uniform / full can be written empty; trained has nothing to train on)
    for storage in ('uniform', 'full'):
        path = tmp_path / ('empty_' + storage + '.bin')
        native.Builder(6, storage, 8).save(path)
        reader = native.Reader(path)
        assert reader.keys() == [] and len(reader) == 0 and reader.dim == 6
        assert bits_equal(reader[['a', '']], np.zeros((2, 6), dtype=np.float32))
        assert bits_equal(reader[['a', '']], oracle.OracleReader(str(path)).batch_embedding(['a', '']))
    with pytest.raises(RuntimeError, match='Nothing to encode'):
        native.Builder(6, 'trained', 4).save(tmp_path / 'empty_trained.bin')


def test_sharded_reader_in_one_process(native, make_model):
    # the node-level path of north_star: one replica per device, host-side gather into one buffer
    # (distinct devices where the box has them; on a one-GPU box the same device listed three times
    # still exercises the split and the threads)
    path, words = make_model(20000, 300, 'trained', 4)
    available = native.hip_device_count()
    devices = [0, 1 % available, 2 % available]
    sharded = native.ShardedReader(path, devices=devices)
    checker = oracle.OracleReader(path)
    batch = sorted(words)[:7001:3] + ['missing'] * 5
    assert sharded.devices == devices and sharded.dim == 300 and len(sharded) == 20000
    assert bits_equal(sharded[batch], checker.batch_embedding(batch))
    assert bits_equal(sharded[batch[:2]], checker.batch_embedding(batch[:2]))   # fewer entries than devices
    assert bits_equal(sharded['missing'], np.zeros(300, dtype=np.float32))
    assert sharded.batch_embedding([]).shape == (0, 300)
    # large enough for every device to search its own slice on the device (3 x 4096 words and more)
    rng = np.random.default_rng(2)
    large = [words[i] for i in rng.integers(0, len(words), size=15000)]
    large[::41] = ['not-here'] * len(large[::41])
    assert bits_equal(sharded[large], checker.batch_embedding(large))
    assert all(reader.host_rows_decoded == 0 for reader in sharded._readers)


def test_tile_geometries_give_identical_rows(native, make_model, monkeypatch):
    # lanes per word (segments of the side index) and waves per block must not change results
    for bits, distribution in ((4, 'normal'), (8, 'student')):
        path, words = make_model(20000, 300, 'trained', bits, distribution=distribution)
        checker = oracle.OracleReader(path)
        rows = np.arange(0, 20000, 3, dtype=np.uint32)
        rows[5::97] = 0xFFFFFFFF
        expected = checker.rows_embedding(rows)
        for lanes, waves, persistent in ((1, 1, 1), (1, 4, 0), (2, 4, 1), (3, 2, 0), (4, 8, 1), (5, 4, 1), (8, 4, 0),
                                         (8, 8, 1), (16, 2, 1), (25, 1, 0), (64, 1, 1)):
            monkeypatch.setenv('MEMB_HIP_LANES', str(lanes))
            monkeypatch.setenv('MEMB_HIP_WAVES', str(waves))
            reader = native.Reader(path)
            reader.set_option('persistent', 2 if persistent else 0)   # 2 = decode_records_persistent whatever the batch size
            assert bits_equal(reader.rows_embedding(rows), expected), (bits, lanes, waves, persistent)
            wide = np.zeros((len(rows), 304), dtype=np.float32)
            reader.batch_embedding_into([reader.keys()[r] if r < 20000 else '?' for r in rows[:500]], wide[:500], 4)
            assert bits_equal(wide[:500, 4:], expected[:500]), (bits, lanes, waves)
            info = reader.info()
            assert info['waves_per_block'] == waves
            assert info['lanes_per_word'] * info['segment_symbols'] >= 300
            assert info['lanes_per_word'] == -(-300 // info['segment_symbols'])


def test_row_layouts_give_identical_rows(native, make_model, monkeypatch):
    # row records (fixed-size regions, record in front of the stream), compact streams + rowMeta records,
    # compact streams + the two index arrays: same bits, host and device buffers, dumps and random rows
    import torch
    for bits, distribution in ((2, 'normal'), (4, 'normal'), (6, 'student'), (8, 'student')):
        path, words = make_model(20000, 300, 'trained', bits, distribution=distribution)
        checker = oracle.OracleReader(path)
        rng = np.random.default_rng(bits)
        rows = rng.integers(0, 20000, size=9000).astype(np.uint32)
        rows[::53] = 0xFFFFFFFF
        expected = checker.rows_embedding(rows)
        dump = np.arange(20000, dtype=np.uint32)
        expected_dump = checker.rows_embedding(dump)
        sizes = {}
        for name, env in (('records', {}), ('rowmeta', {'MEMB_HIP_ROW_RECORDS': '0'}), ('arrays', {'MEMB_HIP_ROW_META': '0'})):
            for key in ('MEMB_HIP_ROW_RECORDS', 'MEMB_HIP_ROW_META'):
                monkeypatch.delenv(key, raising=False)
            for key, value in env.items():
                monkeypatch.setenv(key, value)
            for persistent in ('2', '1', '0'):   # always persistent / by batch size / one tile per wavefront
                reader = native.Reader(path)
                reader.set_option('persistent', int(persistent))
                assert bits_equal(reader.rows_embedding(rows), expected), (bits, name, persistent)
                device_rows = torch.from_numpy(dump.view(np.int32)).cuda()
                assert bits_equal(reader.rows_embedding_device(device_rows).cpu().numpy(), expected_dump), (bits, name, persistent)
                sizes[name] = reader.info()['row_layout']
        # the layouts were really staged; models whose longest row is far above the average row (heavy
        # tails) keep the compact layout by the 35 % padding rule
        assert sizes['rowmeta'] == 1 and sizes['arrays'] == 0, sizes
        assert sizes['records'] == 2 if distribution == 'normal' else sizes['records'] in (1, 2), (bits, sizes)


def test_generic_path_on_a_small_codebook(native, make_model, monkeypatch):
    # <= 16 centroids normally take the nibble / pair-table variant; the byte-key variant must agree
    path, words = make_model(20000, 300, 'trained', 4)
    rows = np.arange(0, 20000, 7, dtype=np.uint32)
    rows[3::50] = 0xFFFFFFFF
    expected = oracle.OracleReader(path).rows_embedding(rows)
    monkeypatch.setenv('MEMB_HIP_NO_FAST', '1')
    assert bits_equal(native.Reader(path).rows_embedding(rows), expected)
    monkeypatch.delenv('MEMB_HIP_NO_FAST')
    assert bits_equal(native.Reader(path).rows_embedding(rows), expected)


def test_randomized_models_and_batches(native, tmp_path, monkeypatch):
    """Seeded sweep: random dimension, vocabulary, storage, bit width, lanes per word, batch make-up,
    output stride / column offset -- HIP path vs CPU checker, bit for bit."""
    # (MEMB_TEST_SWEEP_TRIALS / MEMB_TEST_SWEEP_SEED: longer one-off sweeps with other seeds)
    rng = np.random.default_rng(int(os.environ.get('MEMB_TEST_SWEEP_SEED', 2024)))
    for trial in range(int(os.environ.get('MEMB_TEST_SWEEP_TRIALS', 40))):
        dim = int(rng.choice([1, 2, 3, 4, 7, 8, 12, 16, 20, 31, 32, 48, 63, 64, 96, 100, 128, 200, 257, 300, 512]))
        count = int(rng.integers(1, 2500))
        storage = str(rng.choice(['trained', 'trained', 'trained', 'uniform', 'full']))
        bits = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8]))
        scale = float(rng.choice([1e-3, 0.4, 50.0]))
        if rng.random() < 0.5:
            vectors = (rng.standard_normal((count, dim)) * scale).astype(np.float32)
        else:
            vectors = (rng.standard_t(3, size=(count, dim)) * scale).astype(np.float32)
        words = ['t{}w{}'.format(trial, i) for i in rng.permutation(count)]
        builder = native.Builder(dim, storage, bits)
        builder.add_words(words, vectors)
        path = tmp_path / 'r{}.bin'.format(trial)
        builder.save(path)

        monkeypatch.setenv('MEMB_HIP_LANES', str(int(rng.choice([1, 2, 3, 4, 8, 8, 16, 32]))))
        monkeypatch.setenv('MEMB_HIP_ROOT_BITS', str(int(rng.choice([1, 2, 4, 8, 11, 12]))))
        reader = native.Reader(path)
        reader.set_option('persistent', int(rng.integers(0, 3)))
        checker = oracle.OracleReader(str(path))

        n = int(rng.choice([1, 2, 5, 64, 65, 300, 1500]))
        batch = [words[i] for i in rng.integers(0, count, size=n)]
        miss_rate = float(rng.choice([0.0, 0.1, 0.9]))
        batch = [w if rng.random() >= miss_rate else w + '?' for w in batch]
        expected = checker.batch_embedding(batch)
        assert nan_aware_equal(reader.batch_embedding(batch), expected), (trial, dim, count, storage, bits)
        assert nan_aware_equal(reader.batch_embedding_device(batch).cpu().numpy(), expected), (trial, 'device', dim, storage, bits)

        pad = int(rng.choice([0, 1, 4, 5, 64]))
        col_off = int(rng.choice([0, 1, 4, 8]))
        wide = np.full((n, col_off + dim + pad), 3.25, dtype=np.float32)
        reader.batch_embedding_into(batch, wide, col_off)
        assert nan_aware_equal(wide[:, col_off:col_off + dim], expected), (trial, 'strided', dim, col_off, pad)
        assert (np.delete(wide, np.s_[col_off:col_off + dim], axis=1) == 3.25).all()


def test_concurrent_callers(native, make_model):
    # reference: all read methods are const and re-entrant (src/reader.h:19-27); here the word search is
    # lock-free, the host-buffer entry point serialises per context, the device entry point is per stream
    import threading
    path_a, words_a = make_model(20000, 300, 'trained', 4)
    path_b, words_b = make_model(20000, 300, 'trained', 8, distribution='student')
    readers = [native.Reader(path_a, num_threads=2), native.Reader(path_b, num_threads=2)]
    checkers = [oracle.OracleReader(path_a), oracle.OracleReader(path_b)]
    vocab = [sorted(words_a), sorted(words_b)]
    rng = np.random.default_rng(5)
    jobs = []
    for k in range(12):
        which = k % 2
        size = int(rng.choice([1, 70, 1500, 6000]))
        batch = [vocab[which][i] for i in rng.integers(0, 20000, size=size)] + ['?']
        jobs.append((which, batch, checkers[which].batch_embedding(batch)))
    failures = []

    def run(which, batch, expected):
        for _ in range(3):
            got = readers[which].batch_embedding(batch)
            if not bits_equal(got, expected):
                failures.append((which, len(batch)))

    threads = [threading.Thread(target=run, args=job) for job in jobs]
    for thread in threads:
        thread.start()
    for thread in threads:
        thread.join()
    assert not failures


@pytest.mark.parametrize('bits', [4, 6])
def test_host_results_written_with_non_temporal_stores(native, make_model, monkeypatch, bits):
    """Host buffers: the host threads that expand centroid indices into fp32 rows use non-temporal stores for results of
    64 MB and more (MEMB_HIP_HOST_STREAMING=2: always) when every row is 16-byte aligned, plain stores otherwise --
    the same bits either way, missing rows zero (reference src/reader.cpp:41-47)."""
    path, words = make_model(6000, 300, 'trained', bits)
    checker = oracle.OracleReader(path)
    rng = np.random.default_rng(bits)
    rows = rng.integers(0, len(words), size=9001).astype(np.uint32)
    rows[::37] = 0xFFFFFFFF
    expected = checker.rows_embedding(rows)
    for mode in ('2', '0'):
        monkeypatch.setenv('MEMB_HIP_HOST_STREAMING', mode)
        reader = native.Reader(path, device=0)
        assert bits_equal(reader.rows_embedding(rows), expected), mode
        wide = np.full((len(rows), 304), 7.0, dtype=np.float32)      # aligned rows inside a wider buffer
        reader.rows_embedding_into(rows, wide, 4)
        assert bits_equal(np.ascontiguousarray(wide[:, 4:]), expected) and (wide[:, :4] == 7.0).all(), mode
        odd = np.full((len(rows), 301), 7.0, dtype=np.float32)       # rows that are not 16-byte aligned: plain stores
        reader.rows_embedding_into(rows, odd, 1)
        assert bits_equal(np.ascontiguousarray(odd[:, 1:]), expected) and (odd[:, 0] == 7.0).all(), mode


@pytest.mark.parametrize('bits,distribution', [(4, 'normal'), (2, 'normal'), (6, 'student'), (8, 'student')])
def test_every_kernel_of_a_model_produces_the_same_rows(native, make_model, bits, distribution):
    """The kernels a single trained model can run -- decode_trained with one, two and five tiles per wavefront
    (option 'tiles_per_wave') in blocks of 4, 7 and 8 wavefronts, and decode_records_persistent (option 'persistent' = 2) --
    against the checker: dense rows, strided rows, unaligned output (scalar stores), host batches (centroid indices over
    PCIe) and ragged / tiny batches, with missing rows in every one."""
    import torch
    path, words = make_model(20000, 300, 'trained', bits, distribution=distribution)
    checker = oracle.OracleReader(path)
    reader = native.Reader(path, device=0)
    assert reader.info()['row_layout'] == 2   # row records
    # (the 8-bit model's row regions are too long for a tile of eight to fit the pipeline's two 64-lane rounds of pieces:
    # it runs decode_trained whatever is asked for)
    has_records = bits <= 6
    rng = np.random.default_rng(bits)
    for count in (20000, 4097, 9, 1):
        rows = rng.integers(0, len(words), size=count).astype(np.uint32)
        rows[rng.integers(0, count, size=max(1, count // 50))] = 0xFFFFFFFF
        if count == 20000:
            rows[:12000] = np.arange(12000, dtype=np.uint32)   # a key-order run as well
        expected = checker.rows_embedding(rows)
        ids = torch.from_numpy(rows.view(np.int32)).cuda()
        # (persistent, tiles_per_wave, waves_per_block; seven wavefronts = the blocks of very large batches in no particular order)
        for persistent, steps, waves in ((0, 1, 0), (0, 2, 8), (0, 5, 4), (0, 1, 7), (0, 3, 7), (2, 0, 0), (2, 0, 8), (2, 0, 7), (1, 0, 0)):
            reader.set_option('persistent', persistent)
            reader.set_option('tiles_per_wave', steps)
            reader.set_option('waves_per_block', waves)
            name = reader.info(count)['kernel']
            if persistent != 1:
                assert name.startswith('decode_records_persistent<' if persistent == 2 and has_records else 'decode_trained<'), name
            setting = (persistent, steps, waves, count)
            dense = reader.rows_embedding_device(ids)
            assert bits_equal(dense.cpu().numpy(), expected), setting
            wide = torch.full((count, 640), 7.0, dtype=torch.float32, device='cuda')
            reader.rows_embedding_device(ids, out=wide, col_off=320)
            assert bits_equal(wide[:, 320:620].cpu().numpy(), expected), setting
            assert bool((wide[:, :320] == 7.0).all()) and bool((wide[:, 620:] == 7.0).all())
            odd = torch.zeros((count, 301), dtype=torch.float32, device='cuda')   # rows not 16-byte aligned: scalar stores
            reader.rows_embedding_device(ids, out=odd, col_off=1)
            assert bits_equal(odd[:, 1:].cpu().numpy(), expected), setting
            assert bits_equal(reader.rows_embedding(rows), expected), setting   # host buffers
    reader.set_option('persistent', 1)
    reader.set_option('tiles_per_wave', 0)
    reader.set_option('waves_per_block', 0)


@pytest.mark.parametrize('bits_a,bits_b,seed_b', [(4, 4, 1234), (6, 8, 1234), (2, 4, 1234), (4, 4, 99), (4, 6, 99), (6, 6, 5)])
def test_union_kernels(native, make_model, bits_a, bits_b, seed_b):
    """decode_union_split (two models staged as row records: the wavefront's word slots divided between them; one and
    three tiles per wavefront) and decode_trained_union ('union_split' = 0, or pairs that do not qualify) against numpy
    over the checker's rows: concatenation and average, words missing from either model."""
    import torch
    from memb_amd import _memb
    path_a, words_a = make_model(20000, 300, 'trained', bits_a, seed=1234)
    path_b, words_b = make_model(15000, 300, 'trained', bits_b, seed=seed_b)
    readers = [native.Reader(path_a, device=0), native.Reader(path_b, device=0)]
    checkers = [oracle.OracleReader(path_a), oracle.OracleReader(path_b)]
    # nibble keys need <= 16 centroids and codes of <= 8 bits: the 4-bit model of seed 99 has a rare centroid with a
    # longer code, so (4, 4, 99) and (4, 6, 99) pair a nibble-key model with a byte-key one -- one kernel all the same
    # (the nibble-key model then goes through its byte-key tables)
    formats = {reader.info()['kernel'].split('<')[1].split(',')[2].strip(' >') for reader in readers}
    assert len(formats) == (2 if seed_b == 99 else 1), formats
    rng = np.random.default_rng(bits_a * 10 + bits_b)
    batch = 90001   # the last tile ragged
    rows = []
    for count in (20000, 15000):
        picks = rng.integers(0, count, size=batch).astype(np.uint32)
        picks[rng.random(batch) < 0.2] = 0xFFFFFFFF
        rows.append(picks)
    expected = [checker.rows_embedding(picks) for checker, picks in zip(checkers, rows)]
    ids = [torch.from_numpy(picks.view(np.int32)).cuda() for picks in rows]
    stream = torch.cuda.current_stream().cuda_stream
    for split, steps in ((1, 0), (1, 1), (1, 3), (0, 0)):
        readers[0].set_option('union_split', split)
        readers[0].set_option('tiles_per_wave', steps)
        merged = torch.full((batch, 600), 3.0, dtype=torch.float32, device='cuda')
        assert _memb.union_rows_to_device([r._impl for r in readers], [t.data_ptr() for t in ids], [0, 300], batch,
                                          merged.data_ptr(), 600, stream, False)
        assert bits_equal(merged.cpu().numpy(), np.concatenate(expected, axis=1)), (split, steps)
        ran = readers[0].info()['union_kernel']   # what that launch was
        # (the 8-bit model's row regions are too long for a tile of eight to fit two 64-lane rounds of pieces)
        if split and max(bits_a, bits_b) <= 6 and all(reader.info()['row_layout'] == 2 for reader in readers):
            # row records: also with regions of different sizes (2-bit + 4-bit), byte keys (6-bit + 6-bit) and mixed key formats
            assert ran.startswith('decode_union_split<'), ran
        else:
            assert ran.startswith('decode_trained_union<'), ran
        mean = torch.full((batch, 300), 3.0, dtype=torch.float32, device='cuda')
        assert _memb.union_rows_to_device([r._impl for r in readers], [t.data_ptr() for t in ids], [0, 0], batch,
                                          mean.data_ptr(), 300, stream, True)
        assert bits_equal(mean.cpu().numpy(), np.mean(expected, axis=0)), (split, steps)
    # small and ragged batches through the split kernel (a tile is four words there)
    readers[0].set_option('union_split', 1)
    for steps in (0, 2):
        readers[0].set_option('tiles_per_wave', steps)
        for small in (1, 3, 4, 5, 517):
            merged = torch.full((small, 640), 3.0, dtype=torch.float32, device='cuda')
            assert _memb.union_rows_to_device([r._impl for r in readers], [t[:small].contiguous().data_ptr() for t in ids], [320, 0], small,
                                              merged.data_ptr(), 640, stream, False)
            got = merged.cpu().numpy()
            assert bits_equal(np.ascontiguousarray(got[:, 320:620]), expected[0][:small]), small
            assert bits_equal(np.ascontiguousarray(got[:, 0:300]), expected[1][:small]), small
            assert (got[:, 300:320] == 3.0).all() and (got[:, 620:] == 3.0).all()
    readers[0].set_option('tiles_per_wave', 0)


def test_union_split_with_first_levels_of_different_widths(native, make_model):
    """decode_union_split hands every lane the first-level width of ITS model's table; the second-level index
    must start where that first level ended. Two byte-key models with sub-tables and different widths
    (`max_direct_decode_bits` 5 beside 8, 4 beside 7, and in either order, so both are once the model whose
    slot geometry the kernel runs on) against the checker."""
    import torch
    from memb_amd import _memb
    path_a, words_a = make_model(20000, 300, 'trained', 6, seed=1234, distribution='student')
    path_b, words_b = make_model(15000, 300, 'trained', 6, seed=5, distribution='student')
    checkers = [oracle.OracleReader(path_a), oracle.OracleReader(path_b)]
    rng = np.random.default_rng(610)
    batch = 30011
    rows = []
    for count in (20000, 15000):
        picks = rng.integers(0, count, size=batch).astype(np.uint32)
        picks[rng.random(batch) < 0.1] = 0xFFFFFFFF
        rows.append(picks)
    expected = [checker.rows_embedding(picks) for checker, picks in zip(checkers, rows)]
    ids = [torch.from_numpy(picks.view(np.int32)).cuda() for picks in rows]
    stream = torch.cuda.current_stream().cuda_stream
    for bits_a, bits_b in ((5, 8), (8, 5), (4, 7), (7, 4)):   # (the codes of these models are 9-11 bits long)
        readers = [native.Reader(path_a, device=0, max_direct_decode_bits=bits_a),
                   native.Reader(path_b, device=0, max_direct_decode_bits=bits_b)]
        widths = [reader.info()['root_bits'] for reader in readers]
        assert widths[0] != widths[1], widths
        assert all(reader.info()['max_code_bits'] > width for reader, width in zip(readers, widths))   # sub-tables in both
        for order in ((0, 1), (1, 0)):
            pair = [readers[k] for k in order]
            merged = torch.full((batch, 600), 3.0, dtype=torch.float32, device='cuda')
            assert _memb.union_rows_to_device([r._impl for r in pair], [ids[k].data_ptr() for k in order], [0, 300], batch,
                                              merged.data_ptr(), 600, stream, False)
            assert pair[0].info()['union_kernel'].startswith('decode_union_split<true, false'), pair[0].info()['union_kernel']
            assert bits_equal(merged.cpu().numpy(), np.concatenate([expected[k] for k in order], axis=1)), (bits_a, bits_b, order)
            mean = torch.full((batch, 300), 3.0, dtype=torch.float32, device='cuda')
            assert _memb.union_rows_to_device([r._impl for r in pair], [ids[k].data_ptr() for k in order], [0, 0], batch,
                                              mean.data_ptr(), 300, stream, True)
            assert bits_equal(mean.cpu().numpy(), np.mean([expected[k] for k in order], axis=0)), (bits_a, bits_b, order)


def test_uniform_tile_kernel(native, make_model):
    """dequant_uniform_tile (one tile of eight words per wavefront, row regions through LDS) and the block kernel
    (option 'persistent' = 0; also what unaligned output takes) against the checker: dense, strided, accumulate / divide
    epilogue, unaligned output, missing rows, a ragged last tile, tiny batches; bit-exact as every uniform result."""
    import torch
    path, words = make_model(3000, 300, 'uniform', 8)
    reader = native.Reader(path, device=0)
    checker = oracle.OracleReader(path)
    assert reader.info()['kernel'].startswith('dequant_uniform_tile')
    rng = np.random.default_rng(8)
    for count, block_kernel in ((60001, 0), (36863, 0), (60001, 1), (7, 0), (1, 0)):
        reader.set_option('persistent', 0 if block_kernel else 1)
        assert reader.info()['kernel'].startswith('dequant_uniform<' if block_kernel else 'dequant_uniform_tile<')
        rows = rng.integers(0, len(words), size=count).astype(np.uint32)
        rows[rng.random(count) < 0.03] = 0xFFFFFFFF
        rows[:min(count, 3000)] = np.arange(min(count, 3000), dtype=np.uint32)
        expected = checker.rows_embedding(rows)
        ids = torch.from_numpy(rows.view(np.int32)).cuda()
        assert bits_equal(reader.rows_embedding_device(ids).cpu().numpy(), expected)
        wide = torch.full((count, 640), 7.0, dtype=torch.float32, device='cuda')
        reader.rows_embedding_device(ids, out=wide, col_off=320)
        assert bits_equal(wide[:, 320:620].cpu().numpy(), expected)
        assert bool((wide[:, :320] == 7.0).all()) and bool((wide[:, 620:] == 7.0).all())
        odd = torch.zeros((count, 301), dtype=torch.float32, device='cuda')
        reader.rows_embedding_device(ids, out=odd, col_off=1)
        assert bits_equal(odd[:, 1:].cpu().numpy(), expected)
        accumulated = torch.full((count, 300), 0.5, dtype=torch.float32, device='cuda')
        reader.rows_embedding_device(ids, out=accumulated, accumulate=True, divisor=2.0)
        assert bits_equal(accumulated.cpu().numpy(), (np.float32(0.5) + expected) / np.float32(2.0))
        assert bits_equal(reader.rows_embedding(rows), expected)   # host buffers
    reader.set_option('persistent', 1)


@pytest.mark.parametrize('bits,dim', [(4, 300), (2, 300), (6, 300), (8, 300), (4, 128), (6, 100), (4, 52), (4, 1024), (4, 20)])
def test_small_batches_decode_with_the_finer_index(native, make_model, bits, dim):
    """Row-record models carry a second, finer segment index (about sixteen lanes per word; memb_hip.hip: stageIndex,
    planTrained): batches whose tiles under it fit the CUs at once run decode_trained with it. Same rows as the
    checker with the index forced on (2), off (1) and by rule (0), for batch sizes on both sides of a tile and of the
    rule's edge, misses included. Reference: src/huffman_table_decoder.h:102-118 (the serial chain being split)."""
    import torch
    path, words = make_model(20000, dim, 'trained', bits)
    reader = native.Reader(path)
    checker = oracle.OracleReader(path)
    large, small = reader.info(), reader.info(1000)
    assert small['kernel'].startswith('decode_trained<')
    if dim >= 100:
        # a finer index exists and small batches use it: more lanes per word, fewer symbols per lane
        assert small['lanes_per_word'] > large['lanes_per_word'], (small, large)
        assert small['segment_symbols'] < large['segment_symbols']
    rng = np.random.default_rng(bits * 1000 + dim)
    cus = torch.cuda.get_device_properties(0).multi_processor_count

    def round_of(info):
        # tiles resident at once: per CU what the registers admit (28: memb_hip.hip ONE_TILE_WAVES_PER_CU) or what LDS --
        # 160 KiB handed out per block in steps of 1 KiB -- leaves of it (the 8-bit model's tables, dim 1024's slots)
        waves, lds = info['waves_per_block'], info['lds_bytes_per_block']
        return cus * waves * min(160 * 1024 // ((lds + 1023) // 1024 * 1024), max(1, 28 // waves))

    reader.set_option('fine_lanes', 1)
    usual_round = round_of(reader.info(1000))
    reader.set_option('fine_lanes', 0)
    fine_words, usual_words = 64 // small['lanes_per_word'], 64 // large['lanes_per_word']
    edge = round_of(small) * fine_words                  # the finer index while its tiles fit one round: 28 672 words on 256 CUs
    # (where the usual tiles just miss one round the finer index was the rule for a few hours; it is the pipeline's now)
    island = (usual_round * usual_words, min(2 * 16 * cus, usual_round * 6 // 5) * usual_words)

    def expected_lanes(count):
        return small['lanes_per_word'] if count <= edge else large['lanes_per_word']

    for count in (1, 3, 4, 5, 64, 1000, edge - 1, edge, edge + 9, island[0], island[0] + 1, island[1], island[1] + 1):
        rows = rng.integers(0, len(words), size=count).astype(np.uint32)
        rows[rng.integers(0, count, size=max(1, count // 20))] = 0xFFFFFFFF
        expected = checker.rows_embedding(rows)
        ids = torch.from_numpy(rows.view(np.int32)).cuda()
        for fine in (2, 1, 0):
            reader.set_option('fine_lanes', fine)
            out = torch.full((count, dim), 5.0, dtype=torch.float32, device='cuda')
            reader.rows_embedding_device(ids, out=out)
            torch.cuda.synchronize()
            assert bits_equal(out.cpu().numpy(), expected), (count, fine, reader.info(count)['lanes_per_word'])
            by_rule = reader.info(count)['lanes_per_word']
            if fine == 0 and dim >= 100:
                assert by_rule == expected_lanes(count), (count, by_rule)
        # host buffers take the same path (centroid indices over PCIe for trained storages)
        assert bits_equal(reader.rows_embedding(rows), expected), count
    reader.set_option('fine_lanes', 0)
    assert reader.host_rows_decoded == 0
