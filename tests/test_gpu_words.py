"""Word -> row on the device: memb_hip_ctx_stage_words / memb_hip_words_* /
memb_hip_resolve_rows_device and everything that is fed from them (Reader.resolve_rows_device,
batch_embedding_device, tokenizer_embedding_device, ReadersUnion.batch_embedding_device), against the CPU
checker's search -- the reference's lower_bound + strcmp (src/trained_compression.cpp:115-125) and LookupByKey
(src/uniform_compression.cpp:56, src/full_compression.cpp:39) restated in oracle/memb_oracle.c. Bit-exact: row ids
are integers. Also memb_hip_decode_batches_device (several batches in one launch)."""
import ctypes
import os

import numpy as np
import pytest

import oracle
from conftest import bits_equal

pytestmark = pytest.mark.gpu

FULL_VOCAB = 2196017
MISSING = 0xFFFFFFFF

# keys that a hash + compare can get wrong where a binary search cannot: the empty word, prefixes of one another,
# bytes above 0x7F (strcmp compares unsigned bytes; UTF-8 of two, three and four bytes), a word longer than a
# wavefront's LDS stage, neighbours that differ in the last byte only
TRICKY_KEYS = ['', 'a', 'ab', 'abc', 'abcd', 'th', 'the', 'then', 'theng', 'naïve', 'nai', 'naive',
               '日本語', '日本', '日', '\U0001f600', '\U0001f601', 'zz', 'zzz',
               'x' * 3000, 'x' * 2999 + 'y', 'q' * 40, 'q' * 41]


def tricky_queries(keys, rng):
    queries = list(keys)
    queries += [key + 'z' for key in keys[:40]]                 # a key is a proper prefix of the query
    queries += [key[:-1] for key in keys[:40] if key]           # the query is a proper prefix of a key
    queries += ['\x01', '\x7f', 'ÿ' * 3, '\U0010ffff', '~~~~', ' ']   # sort before the first / after the last key
    queries += ['the\x00n', 'abc\x00', '\x00abc']               # a word ends at its first NUL (strcmp)
    queries += ['y' * 5000, 'x' * 3001, 'x' * 2998]             # longer than the LDS stage of a wavefront
    queries += [queries[i] for i in rng.integers(0, len(queries), size=300)]   # repeats
    order = rng.permutation(len(queries))
    return [queries[i] for i in order]


@pytest.fixture(scope='module')
def tricky_models(native, tmp_path_factory):
    """One small model per storage over TRICKY_KEYS + 700 synthetic words (several wavefronts of keys)."""
    from memb_amd import synthetic
    directory = tmp_path_factory.mktemp('words')
    keys = TRICKY_KEYS + synthetic.make_words(700, seed=3)
    rng = np.random.default_rng(5)
    vectors = rng.standard_normal((len(keys), 20)).astype(np.float32)
    paths = {}
    for storage, bits in (('trained', 4), ('uniform', 8), ('full', 32)):
        builder = native.Builder(20, storage, bits)
        order = rng.permutation(len(keys))       # insertion order is not key order
        builder.add_words([keys[i] for i in order], vectors[order])
        paths[storage] = str(directory / (storage + '.bin'))
        builder.save(paths[storage])
    return paths, keys


@pytest.mark.parametrize('storage', ['trained', 'uniform', 'full'])
def test_tricky_words_on_every_storage(native, tricky_models, storage):
    import torch
    paths, keys = tricky_models
    reader = native.Reader(paths[storage])
    checker = oracle.OracleReader(paths[storage])
    queries = tricky_queries(keys, np.random.default_rng(1))
    expected = checker.resolve_rows(queries)
    assert (expected != MISSING).sum() >= len(keys) and (expected == MISSING).sum() > 50
    rows = reader.resolve_rows_device(queries)
    assert rows.dtype == torch.int32 and rows.device.type == 'cuda'
    got = rows.cpu().numpy().view(np.uint32)
    assert np.array_equal(got, expected), [(queries[i][:20], got[i], expected[i]) for i in np.nonzero(got != expected)[0][:5]]
    assert np.array_equal(reader.resolve_rows(queries), expected)        # the host search gives the same
    info = reader.info()
    assert info['word_index_keys'] == len(keys) and info['word_index_slots'] >= 2 * len(keys)
    assert 0 < info['word_index_bytes'] <= info['device_bytes']
    # the rows feed the decode without a host round trip
    assert bits_equal(reader.batch_embedding_device(queries).cpu().numpy(), checker.batch_embedding(
        [q.split('\x00')[0] for q in queries]))
    # the reference's own API -- words in, numpy out -- searches on the device too from 4096 words on
    many = queries * (4096 // len(queries) + 2)
    assert len(many) >= 4096
    expected_many = checker.batch_embedding([q.split('\x00')[0] for q in many])
    assert bits_equal(reader[many], expected_many)
    wide = np.full((len(many), 50), 3.0, dtype=np.float32)
    reader.batch_embedding_into(many, wide, 30)
    assert bits_equal(wide[:, 30:], expected_many) and (wide[:, :30] == 3.0).all()
    assert reader.host_rows_decoded == 0
    # edge sizes: nothing, one word, one wavefront +- 1, one block +- 1
    for count in (0, 1, 63, 64, 65, 255, 256, 257):
        part = queries[:count]
        assert np.array_equal(reader.resolve_rows_device(part).cpu().numpy().view(np.uint32), expected[:count]), count
    with pytest.raises(TypeError):
        reader.resolve_rows_device(['a', 7])
    with pytest.raises(TypeError):
        reader.resolve_rows_device(['a'], out=torch.empty(2, dtype=torch.int32, device='cuda'))


@pytest.fixture(scope='module')
def full_model(native):
    from memb_amd import synthetic
    count = int(os.environ.get('MEMB_TEST_FULL_VOCAB', FULL_VOCAB))
    path, _ = synthetic.cached_model(count, 300, 'trained', 4)   # shared with bench.py and test_gpu_full_size.py
    return path, count


def test_full_vocabulary_word_search(native, full_model):
    """The 2.2 M-key model: all keys in key order, all keys shuffled, misses on both ends and in between, repeats."""
    import torch
    path, count = full_model
    reader = native.Reader(path)
    checker = oracle.OracleReader(path)
    keys = reader.keys()
    assert len(keys) == count
    rows = reader.resolve_rows_device(keys)
    assert torch.equal(rows, torch.arange(count, dtype=torch.int32, device='cuda'))
    assert reader.info()['word_index_keys'] == count

    rng = np.random.default_rng(23)
    order = rng.permutation(count)
    shuffled = [keys[i] for i in order]
    got = reader.resolve_rows_device(shuffled).cpu().numpy().view(np.uint32)
    assert np.array_equal(got, order.astype(np.uint32))

    sample = rng.integers(0, count, size=300000)
    mixed = [keys[i] for i in sample]
    for i in range(0, len(mixed), 7):
        mixed[i] = mixed[i] + '!'          # between keys
    for i in range(3, len(mixed), 1001):
        mixed[i] = '\x01' + mixed[i]       # before the first key
    for i in range(5, len(mixed), 1003):
        mixed[i] = '\U0010ffff' + mixed[i]  # after the last key: the reference's end() dereference
    mixed[11] = ''
    expected = checker.resolve_rows(mixed)
    assert (expected == MISSING).sum() > 40000
    got = reader.resolve_rows_device(mixed).cpu().numpy().view(np.uint32)
    assert np.array_equal(got, expected)
    assert np.array_equal(reader.resolve_rows(mixed), expected)
    # the device API end to end on the model of BASELINE.json configs[1]
    out = reader.batch_embedding_device(mixed[:50000])
    assert bits_equal(out.cpu().numpy(), checker.rows_embedding(expected[:50000]))
    assert reader.host_rows_decoded == 0


def test_batch_sizes_around_every_packing_boundary(native, make_model):
    """The packer splits a batch into jobs of 16 384 words and the jobs into copy chunks: sizes on both sides of every
    boundary (round 5's first version waited forever on a chunk without jobs at 19 jobs)."""
    path, words = make_model(6000, 300, 'trained', 4)
    reader = native.Reader(path)
    rng = np.random.default_rng(7)
    # (fresh non-ASCII str objects have no cached UTF-8 form: the first -- pooled -- fill meets them, has the calling thread
    # prepare them through the API and fills again)
    pool = words + ['miss-{}'.format(i) for i in range(600)] + ['fehlt-ä{}ß'.format(i) for i in range(200)] + ['没有{}'.format(i) for i in range(100)]
    longest = [pool[i] for i in rng.integers(0, len(pool), size=25 * 16384 + 1)]
    expected = reader.resolve_rows(longest)
    assert (expected == MISSING).sum() > 1000
    sizes = [16383, 16384, 16385, 2 * 16384 + 1, 100000]
    sizes += [jobs * 16384 - delta for jobs in range(3, 26) for delta in (0, 16383)]
    for size in sorted(set(sizes)) + [len(longest)]:
        got = reader.resolve_rows_device(longest[:size]).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, expected[:size]), size


@pytest.mark.parametrize('seed', [1, 2, 3])
def test_random_vocabularies_of_look_alikes(native, tmp_path, seed):
    """Property-style: vocabularies drawn from a tiny alphabet (so that keys share long prefixes, differ in one byte, are
    prefixes of one another) mixed with multi-byte UTF-8, lengths 0 .. 40; queried with every key, with mutations of keys
    (a byte changed, dropped, appended) and with fresh draws from the same generator -- device search == checker's binary
    search == host search on every word, for a trained and a uniform storage."""
    rng = np.random.default_rng(seed)
    alphabet = ['a', 'b', 'ab', 'ba', 'é', 'ß', '字', '\U0001f600', '-', 'aa']

    def draw(count):
        lengths = rng.integers(0, 41, size=count)
        return [''.join(alphabet[i] for i in rng.integers(0, len(alphabet), size=length)) for length in lengths]

    keys = sorted(set(draw(6000)), key=lambda k: k.encode())
    vectors = rng.standard_normal((len(keys), 8)).astype(np.float32)
    queries = list(keys)
    for key in keys[::3]:
        if key:
            position = int(rng.integers(0, len(key)))
            queries.append(key[:position] + 'b' + key[position + 1:])   # one character changed
            queries.append(key[:position] + key[position + 1:])        # one dropped
        queries.append(key + alphabet[int(rng.integers(0, len(alphabet)))])   # one appended
    queries += draw(5000)
    order = rng.permutation(len(queries))
    queries = [queries[i] for i in order]
    for storage, bits in (('trained', 4), ('uniform', 8)):
        path = str(tmp_path / '{}_{}.bin'.format(storage, seed))
        builder = native.Builder(8, storage, bits)
        insertion = rng.permutation(len(keys))
        builder.add_words([keys[i] for i in insertion], vectors[insertion])
        builder.save(path)
        reader = native.Reader(path)
        checker = oracle.OracleReader(path)
        assert reader.keys() == keys
        expected = checker.resolve_rows(queries)
        hits = int((expected != MISSING).sum())
        assert hits >= len(keys) and hits < len(queries) - 1000
        got = reader.resolve_rows_device(queries).cpu().numpy().view(np.uint32)
        wrong = np.nonzero(got != expected)[0]
        assert len(wrong) == 0, [(queries[i], int(got[i]), int(expected[i])) for i in wrong[:5]]
        assert np.array_equal(reader.resolve_rows(queries), expected)
        assert bits_equal(reader[queries], checker.batch_embedding(queries))   # (>= 4096 words: the device search inside)


def _library(native):
    library = ctypes.CDLL(native.HIP_LIBRARY_PATH)
    library.memb_hip_last_error.restype = ctypes.c_char_p
    return library


class UniformRow(ctypes.Structure):
    _fields_ = [('values', ctypes.c_void_p), ('n_values', ctypes.c_uint32),
                ('min_value', ctypes.c_float), ('max_value', ctypes.c_float)]


class UniformDesc(ctypes.Structure):
    _fields_ = [('dim', ctypes.c_uint32), ('n_rows', ctypes.c_uint64), ('rows', ctypes.c_void_p),
                ('quantization_levels', ctypes.c_uint8)]


def _uniform_context(library, count, dim=4):
    payload = np.arange(count * dim, dtype=np.uint8)
    rows = (UniformRow * count)()
    for i in range(count):
        rows[i] = UniformRow(payload.ctypes.data + i * dim, dim, 0.0, 1.0)
    desc = UniformDesc(dim, count, ctypes.addressof(rows), 255)
    context = ctypes.c_void_p()
    assert library.memb_hip_ctx_create_uniform(ctypes.byref(context), 0, ctypes.byref(desc)) == 0, library.memb_hip_last_error()
    return context


def test_word_search_through_the_c_abi(native):
    """stage_words / words_pack / resolve_rows_device / resolve_packed_device as a foreign caller binds them; a file
    with REPEATED keys resolves to the first of them, as lower_bound does; the refusals."""
    import torch
    library = _library(native)
    keys = [b'a', b'b', b'b', b'b', b'c', b'cc', b'cc', b'd']       # sorted, with repeats
    packed = b''.join(key + b'\x00' for key in keys)
    offsets = np.cumsum([0] + [len(key) + 1 for key in keys[:-1]]).astype(np.uint32)
    context = _uniform_context(library, len(keys))
    stage = library.memb_hip_ctx_stage_words
    stage.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64]
    # refusals first: a key count that is not the row count, keys without the final NUL, an offset outside
    assert stage(context, packed, len(packed), offsets.ctypes.data, len(keys) - 1) == 1
    assert stage(context, packed[:-1], len(packed) - 1, offsets.ctypes.data, len(keys)) == 1
    bad = offsets.copy()
    bad[3] = len(packed)
    assert stage(context, packed, len(packed), bad.ctypes.data, len(keys)) == 1
    batch = ctypes.c_void_p()
    assert library.memb_hip_words_create(ctypes.byref(batch), 0) == 0
    rows = torch.full((16,), 7, dtype=torch.int32, device='cuda')
    resolve = library.memb_hip_resolve_rows_device
    resolve.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    queries = [b'b', b'cc', b'a', b'd', b'', b'bb', b'c', b'e', b'cc\x00x']
    array = (ctypes.c_char_p * len(queries))(*queries)
    pack = library.memb_hip_words_pack
    pack.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    assert resolve(context, batch, rows.data_ptr(), None) == 1        # nothing committed yet
    assert pack(batch, array, None, len(queries)) == 0, library.memb_hip_last_error()
    assert resolve(context, batch, rows.data_ptr(), None) == 1        # keys not staged yet
    assert b'stage_words' in library.memb_hip_last_error()
    assert stage(context, packed, len(packed), offsets.ctypes.data, len(keys)) == 0, library.memb_hip_last_error()
    assert stage(context, packed, len(packed), offsets.ctypes.data, len(keys)) == 0   # idempotent
    assert resolve(context, batch, rows.data_ptr(), None) == 0, library.memb_hip_last_error()
    torch.cuda.synchronize()
    assert rows.cpu().numpy().view(np.uint32).tolist()[:9] == [1, 5, 0, 7, MISSING, MISSING, 4, MISSING, 5]
    assert rows[9:].eq(7).all()
    count = ctypes.c_size_t()
    assert library.memb_hip_words_count(batch, ctypes.byref(count)) == 0 and count.value == len(queries)
    # explicit lengths (no terminators needed), and a second pack into the same object
    joined = b'ccdab'
    buffer = ctypes.create_string_buffer(joined, len(joined))
    base = ctypes.addressof(buffer)
    pointers = (ctypes.c_void_p * 4)(base, base + 2, base + 3, base + 4)
    lengths = np.array([2, 1, 1, 1], dtype=np.uint32)
    assert pack(batch, pointers, lengths.ctypes.data, 4) == 0, library.memb_hip_last_error()
    assert resolve(context, batch, rows.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert rows.cpu().numpy().view(np.uint32).tolist()[:4] == [5, 7, 0, 1]

    # a caller-filled batch (begin / write the jobs / commit), looked up job range by job range before the commit
    class Plan(ctypes.Structure):
        _fields_ = [('bytes', ctypes.c_void_p), ('offsets', ctypes.c_void_p), ('n', ctypes.c_size_t),
                    ('job_words', ctypes.c_size_t), ('jobs', ctypes.c_size_t), ('job_bytes', ctypes.c_size_t)]
    begin = library.memb_hip_words_begin
    begin.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p]
    plan = Plan()
    many = [b'cc', b'nope', b'a', b'', b'd', b'b'] * 100
    assert begin(batch, len(many), 0, ctypes.byref(plan)) == 0, library.memb_hip_last_error()
    assert plan.n == len(many) and plan.job_words % 64 == 0 and plan.jobs == -(-len(many) // plan.job_words) and plan.jobs > 2
    assert plan.job_bytes % 16 == 0 and plan.job_bytes >= plan.job_words * 4
    host_bytes = (ctypes.c_uint8 * (plan.jobs * plan.job_bytes)).from_address(plan.bytes)
    host_offsets = (ctypes.c_uint32 * (plan.jobs * (plan.job_words + 1))).from_address(plan.offsets)
    for job in range(plan.jobs):
        at = job * plan.job_bytes
        mine = many[job * plan.job_words:(job + 1) * plan.job_words]
        for k, word in enumerate(mine):
            host_offsets[job * (plan.job_words + 1) + k] = at
            host_bytes[at:at + len(word)] = word
            at += len(word)
        host_offsets[job * (plan.job_words + 1) + len(mine)] = at
    big = torch.full((len(many),), 7, dtype=torch.int32, device='cuda')
    in_range = library.memb_hip_resolve_range_device
    in_range.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
    assert resolve(context, batch, big.data_ptr(), None) == 1          # not committed
    assert in_range(context, batch, 1, 10, big.data_ptr(), None) == 1  # not the start of a job
    assert in_range(context, batch, plan.job_words, len(many), big.data_ptr(), None) == 1   # past the end
    assert in_range(context, batch, plan.job_words, plan.job_words, big.data_ptr(), None) == 0, library.memb_hip_last_error()
    torch.cuda.synchronize()
    want = np.array([{b'cc': 5, b'a': 0, b'd': 7, b'b': 1}.get(w, MISSING) for w in many], dtype=np.uint32)
    got = big.cpu().numpy().view(np.uint32)
    assert np.array_equal(got[plan.job_words:2 * plan.job_words], want[plan.job_words:2 * plan.job_words])
    assert (got[:plan.job_words] == 7).all() and (got[2 * plan.job_words:] == 7).all()
    assert library.memb_hip_words_commit(batch) == 0
    assert resolve(context, batch, big.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(big.cpu().numpy().view(np.uint32), want)

    # words that are on the device already; an UNALIGNED byte pointer takes the path without the LDS stage
    packed_device = library.memb_hip_resolve_packed_device
    packed_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
    words = [b'cc', b'zebra', b'', b'b', b'a' * 100, b'd'] * 50
    data = np.frombuffer(b''.join(words), dtype=np.uint8)
    starts = np.cumsum([0] + [len(w) for w in words]).astype(np.uint32)
    expected = [{b'cc': 5, b'b': 1, b'd': 7}.get(w, MISSING) for w in words]
    for shift in (0, 1):
        holder = torch.zeros(len(data) + 32, dtype=torch.uint8, device='cuda')
        holder[shift:shift + len(data)] = torch.from_numpy(data.copy()).cuda()
        device_offsets = torch.from_numpy(starts.view(np.int32).copy()).cuda()
        out = torch.empty(len(words), dtype=torch.int32, device='cuda')
        assert packed_device(context, holder.data_ptr() + shift, device_offsets.data_ptr(), len(words), out.data_ptr(), None) == 0
        torch.cuda.synchronize()
        assert out.cpu().numpy().view(np.uint32).tolist() == expected, shift

    # one launch for two contexts (a ReadersUnion's readers): every word hashed once, probed in each model's table
    other_keys = [b'', b'a', b'b', b'bb', b'nope', b'x', b'y', b'z']
    other_packed = b''.join(key + b'\x00' for key in other_keys)
    other_offsets = np.cumsum([0] + [len(key) + 1 for key in other_keys[:-1]]).astype(np.uint32)
    context2 = _uniform_context(library, len(other_keys))
    assert stage(context2, other_packed, len(other_packed), other_offsets.ctypes.data, len(other_keys)) == 0, library.memb_hip_last_error()
    union_range = library.memb_hip_resolve_range_union_device
    union_range.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
    rows_a = torch.full((len(many),), 7, dtype=torch.int32, device='cuda')
    rows_b = torch.full((len(many),), 7, dtype=torch.int32, device='cuda')
    contexts = (ctypes.c_void_p * 2)(context.value, context2.value)
    targets = (ctypes.c_void_p * 2)(rows_a.data_ptr(), rows_b.data_ptr())
    assert union_range(contexts, 2, batch, 0, len(many), targets, None) == 0, library.memb_hip_last_error()
    torch.cuda.synchronize()
    assert np.array_equal(rows_a.cpu().numpy().view(np.uint32), want)
    want_b = np.array([{b'': 0, b'a': 1, b'b': 2, b'nope': 4}.get(w, MISSING) for w in many], dtype=np.uint32)
    assert np.array_equal(rows_b.cpu().numpy().view(np.uint32), want_b)
    assert union_range(contexts, 5, batch, 0, len(many), targets, None) == 1   # at most four contexts per launch
    library.memb_hip_ctx_destroy(context2)

    # words in, host rows out (memb_hip_decode_words): the same rows as memb_hip_decode_rows gives for the looked-up ids
    decode_words = library.memb_hip_decode_words
    decode_words.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t]
    decode_rows = library.memb_hip_decode_rows
    decode_rows.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t]
    from_words = np.full((len(many), 6), 9.0, dtype=np.float32)
    assert decode_words(context, batch, from_words.ctypes.data, 6, 1) == 0, library.memb_hip_last_error()
    from_rows = np.full((len(many), 6), 9.0, dtype=np.float32)
    assert decode_rows(context, want.ctypes.data, len(many), from_rows.ctypes.data, 6, 1) == 0, library.memb_hip_last_error()
    assert np.array_equal(from_words.view(np.uint32), from_rows.view(np.uint32))
    assert not from_words[want == MISSING, 1:5].any() and (from_words[:, 0] == 9.0).all() and (from_words[:, 5] == 9.0).all()
    assert decode_words(context, batch, from_words.ctypes.data, 3, 0) == 1   # ld < dim

    class Info(ctypes.Structure):
        _fields_ = [('struct_size', ctypes.c_uint32), ('device', ctypes.c_int32), ('storage', ctypes.c_uint32),
                    ('dim', ctypes.c_uint32), ('n_rows', ctypes.c_uint64), ('device_bytes', ctypes.c_uint64),
                    ('trained', ctypes.c_uint32 * 8), ('kernel', ctypes.c_char * 96), ('row_layout', ctypes.c_uint32),
                    ('row_bytes', ctypes.c_uint32), ('kernel_registers', ctypes.c_uint32),
                    ('register_waves_per_cu', ctypes.c_uint32), ('batch_words', ctypes.c_uint64),
                    ('tiles_per_wavefront', ctypes.c_uint32), ('reserved_abi3', ctypes.c_uint32 * 2),
                    ('union_kernel', ctypes.c_char * 96), ('word_index_bytes', ctypes.c_uint64),
                    ('word_index_slots', ctypes.c_uint32), ('word_index_keys', ctypes.c_uint32)]
    info = Info()
    info.struct_size = ctypes.sizeof(Info)
    assert library.memb_hip_ctx_get_info(context, ctypes.byref(info)) == 0, library.memb_hip_last_error()
    assert info.word_index_keys == 5 and info.word_index_slots == 16 and info.word_index_bytes > 0   # a, b, c, cc, d
    assert info.struct_size == ctypes.sizeof(Info) and info.n_rows == len(keys)
    # the ABI-4 declaration of this struct (union_kernel 8 bytes lower) is recognised by its size and refused
    info.struct_size = ctypes.sizeof(Info) - 16 - 8
    assert library.memb_hip_ctx_get_info(context, ctypes.byref(info)) == 1
    assert b'ABI-4' in library.memb_hip_last_error()
    library.memb_hip_words_destroy(batch)
    library.memb_hip_ctx_destroy(context)


def test_union_resolves_one_batch_against_every_reader(native, make_model):
    import torch
    path_a, words_a = make_model(6000, 300, 'trained', 4, seed=1)
    path_b, words_b = make_model(5000, 300, 'trained', 4, seed=2)
    # (both vocabularies come from one word generator: b knows the first 5000 of a's 6000 words and misses the rest)
    readers = [native.Reader(path_a), native.Reader(path_b)]
    checkers = [oracle.OracleReader(path_a), oracle.OracleReader(path_b)]
    rng = np.random.default_rng(3)
    batch = [words_a[i] for i in rng.integers(0, len(words_a), size=9000)]
    batch += ['unknown-{}'.format(i) for i in range(500)] + ['']
    for mode in ('concatenate', 'average'):
        union = native.ReadersUnion(readers, mode)
        out = union.batch_embedding_device(batch)
        pieces = [checker.batch_embedding(batch) for checker in checkers]
        expected = np.concatenate(pieces, axis=-1) if mode == 'concatenate' else np.mean(pieces, axis=0)
        assert bits_equal(out.cpu().numpy(), expected), mode
        assert bits_equal(union.batch_embedding(batch), expected), mode
    assert all(reader.host_rows_decoded == 0 for reader in readers)


def test_tokenizer_embedding_stays_on_the_device(native, make_model):
    path, words = make_model(3000, 300, 'trained', 4)
    reader = native.Reader(path)

    class Tokenizer:
        word_index = {word: index + 1 for index, word in enumerate(words[:500] + ['not-a-word'])}
        num_words = 400
    weights = reader.tokenizer_embedding_device(Tokenizer())
    assert bits_equal(weights.cpu().numpy(), reader.tokenizer_embedding(Tokenizer()))
    assert weights.shape == (400, 300) and not weights[0].any()


@pytest.mark.parametrize('storage,bits', [('trained', 4), ('trained', 6), ('trained', 8), ('uniform', 8), ('full', 32)])
def test_several_batches_in_one_launch(native, make_model, storage, bits):
    """memb_hip_decode_batches_device: K in {1, 3, 8} ragged batches with misses, dense and strided outputs, against the
    checker and against one launch per batch."""
    import torch
    path, words = make_model(4000, 300, storage, bits)
    reader = native.Reader(path)
    checker = oracle.OracleReader(path)
    rng = np.random.default_rng(31)
    # (trained storages: small lists decode with the finer segment index, as one batch of as many words would -- the last
    # list is past that rule on 256 CUs; every list also with the index forced off and on)
    cases = [(sizes, 0) for sizes in ((1000,), (1, 777, 8), (513, 1, 64, 4099, 7, 8, 250, 1031), (20000, 15000, 3))]
    if storage == 'trained':
        cases += [((1, 777, 8), 1), ((513, 1, 64, 4099, 7, 8, 250, 1031), 1), ((20000, 15000, 3), 2)]
    for sizes, fine_lanes in cases:
        if storage == 'trained':
            reader.set_option('fine_lanes', fine_lanes)
        entries = []
        host_rows = []
        for k, size in enumerate(sizes):
            rows = rng.integers(0, len(words), size=size).astype(np.uint32)
            rows[rng.integers(0, size, size=max(1, size // 10))] = MISSING
            host_rows.append(rows)
            ids = torch.from_numpy(rows.view(np.int32)).cuda()
            if k % 3 == 2:      # a column block of a wider matrix
                out = torch.full((size, 640), 3.0, dtype=torch.float32, device='cuda')
                entries.append((ids, out, 320))
            else:
                entries.append((ids, torch.full((size, 300), 3.0, dtype=torch.float32, device='cuda')))
        outs = reader.rows_embedding_device_many(entries)
        torch.cuda.synchronize()
        for k, (rows, out) in enumerate(zip(host_rows, outs)):
            expected = checker.rows_embedding(rows)
            if k % 3 == 2:
                assert bits_equal(out[:, 320:620].cpu().numpy(), expected), (sizes, k, fine_lanes)
                assert out[:, :320].eq(3.0).all() and out[:, 620:].eq(3.0).all()
            else:
                assert bits_equal(out.cpu().numpy(), expected), (sizes, k, fine_lanes)
    # an empty batch among the others, and an output whose rows are not 16-byte aligned (dim 300, ld 301)
    ids = torch.from_numpy(host_rows[0].view(np.int32)).cuda()
    odd = torch.zeros((len(host_rows[0]), 301), dtype=torch.float32, device='cuda')
    empty = (torch.empty(0, dtype=torch.int32, device='cuda'), torch.empty((0, 300), dtype=torch.float32, device='cuda'))
    dense = torch.zeros((len(host_rows[0]), 300), dtype=torch.float32, device='cuda')
    reader.rows_embedding_device_many([(ids, odd), empty, (ids, dense)])
    torch.cuda.synchronize()
    expected = checker.rows_embedding(host_rows[0])
    assert bits_equal(odd[:, :300].cpu().numpy(), expected) and bits_equal(dense.cpu().numpy(), expected)
    with pytest.raises(ValueError):
        reader.rows_embedding_device_many([(ids, torch.zeros((len(host_rows[0]), 200), dtype=torch.float32, device='cuda'))])
    assert reader.host_rows_decoded == 0


def test_device_lookups_capture_into_a_hip_graph(native, make_model):
    """The device APIs only enqueue kernels on the caller's stream once a context has been used (first use stages the model
    and raises the kernels' LDS limit): a lookup, and several lookups in one launch, can be captured into a HIP graph and
    replayed -- with new row ids in the same buffers -- bit for bit like eager launches (round 5, batch 20, profiles/r05_experiments.txt, timed it:
    a replay saves nothing over an eager launch, the one launch for K lookups does)."""
    import torch
    path, words = make_model(6000, 300, 'trained', 4)
    reader = native.Reader(path)
    checker = oracle.OracleReader(path)
    rng = np.random.default_rng(77)
    sizes = (1000, 37, 2500)

    def fresh_rows():
        rows = [rng.integers(0, len(words), size=size).astype(np.uint32) for size in sizes]
        for r in rows:
            r[rng.integers(0, len(r), size=max(1, len(r) // 20))] = MISSING
        return rows

    host_rows = fresh_rows()
    ids = [torch.from_numpy(r.view(np.int32)).cuda() for r in host_rows]
    single = [torch.zeros((size, 300), dtype=torch.float32, device='cuda') for size in sizes]
    many = [torch.zeros((size, 300), dtype=torch.float32, device='cuda') for size in sizes]
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                     # warm-up on the capture stream, outside the capture
        for i in range(len(sizes)):
            reader.rows_embedding_device(ids[i], out=single[i])
        reader.rows_embedding_device_many(list(zip(ids, many)))
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for i in range(len(sizes)):
            reader.rows_embedding_device(ids[i], out=single[i])
        reader.rows_embedding_device_many(list(zip(ids, many)))
    for attempt in range(3):
        if attempt:
            host_rows = fresh_rows()
            for i, r in enumerate(host_rows):
                ids[i].copy_(torch.from_numpy(r.view(np.int32)))
        for t in single + many:
            t.fill_(9.0)
        graph.replay()
        torch.cuda.synchronize()
        for i, r in enumerate(host_rows):
            expected = checker.rows_embedding(r)
            assert bits_equal(single[i].cpu().numpy(), expected), (attempt, i)
            assert bits_equal(many[i].cpu().numpy(), expected), (attempt, i)
    assert reader.host_rows_decoded == 0


@pytest.mark.parametrize('storage', ['uniform', 'full'])
def test_full_vocabulary_word_search_on_node_keyed_storages(native, storage, tmp_path_factory):
    """The uniform and the full storage keep their keys in the nodes of a sorted vector (LookupByKey,
    src/uniform_compression.cpp:56, src/full_compression.cpp:39): the hash table is staged from keys collected node by
    node. 2.2 M keys (narrow vectors: the search does not look at them), key order, shuffled, misses on both ends."""
    import torch
    from memb_amd import synthetic
    count = int(os.environ.get('MEMB_TEST_FULL_VOCAB', FULL_VOCAB))
    path = str(tmp_path_factory.mktemp('wide') / (storage + '.bin'))
    synthetic.build_file(path, count, 4, storage, 8 if storage == 'uniform' else 32, seed=31)
    reader = native.Reader(path)
    checker = oracle.OracleReader(path)
    keys = reader.keys()
    assert len(keys) == count
    assert torch.equal(reader.resolve_rows_device(keys), torch.arange(count, dtype=torch.int32, device='cuda'))
    info = reader.info()
    assert info['word_index_keys'] == count and info['word_index_slots'] >= 2 * count
    rng = np.random.default_rng(29)
    order = rng.permutation(count)
    shuffled = [keys[i] for i in order]
    assert np.array_equal(reader.resolve_rows_device(shuffled).cpu().numpy().view(np.uint32), order.astype(np.uint32))
    mixed = [keys[i] for i in rng.integers(0, count, size=200000)]
    for i in range(0, len(mixed), 5):
        mixed[i] = mixed[i] + '#'
    for i in range(2, len(mixed), 997):
        mixed[i] = '\x01' + mixed[i]
    for i in range(4, len(mixed), 991):
        mixed[i] = '\U0010ffff' + mixed[i]
    expected = checker.resolve_rows(mixed)
    assert (expected == MISSING).sum() > 40000
    assert np.array_equal(reader.resolve_rows_device(mixed).cpu().numpy().view(np.uint32), expected)
    assert bits_equal(reader.batch_embedding_device(mixed[:5000]).cpu().numpy(), checker.batch_embedding(mixed[:5000]))


def test_poisoned_offsets_of_a_caller_filled_batch(native):
    """A caller that fills the pinned job regions itself (memb_hip_words_begin / _commit) and gets offsets wrong: entries
    that run backwards, point past the job regions, or are 0xFFFFFFFF make THEIR words MISSING (resolve_words' `sane` guard);
    every other word of the batch keeps its answer and nothing faults."""
    import torch
    library = _library(native)
    keys = [b'a', b'b', b'c', b'cc', b'd', b'e']
    packed = b''.join(key + b'\x00' for key in keys)
    offsets = np.cumsum([0] + [len(key) + 1 for key in keys[:-1]]).astype(np.uint32)
    context = _uniform_context(library, len(keys))
    stage = library.memb_hip_ctx_stage_words
    stage.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64]
    assert stage(context, packed, len(packed), offsets.ctypes.data, len(keys)) == 0, library.memb_hip_last_error()
    batch = ctypes.c_void_p()
    assert library.memb_hip_words_create(ctypes.byref(batch), 0) == 0

    class Plan(ctypes.Structure):
        _fields_ = [('bytes', ctypes.c_void_p), ('offsets', ctypes.c_void_p), ('n', ctypes.c_size_t),
                    ('job_words', ctypes.c_size_t), ('jobs', ctypes.c_size_t), ('job_bytes', ctypes.c_size_t)]
    begin = library.memb_hip_words_begin
    begin.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p]
    resolve = library.memb_hip_resolve_rows_device
    resolve.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    plan = Plan()
    words = [b'cc', b'a', b'zz', b'd', b'b', b'e', b'c'] * 120    # 840 words: several jobs, several wavefronts
    assert begin(batch, len(words), 0, ctypes.byref(plan)) == 0, library.memb_hip_last_error()
    host_bytes = (ctypes.c_uint8 * (plan.jobs * plan.job_bytes)).from_address(plan.bytes)
    host_offsets = (ctypes.c_uint32 * (plan.jobs * (plan.job_words + 1))).from_address(plan.offsets)

    def entry(word):   # where word's offset lives (every job has one entry more than words)
        return word + word // plan.job_words

    for job in range(plan.jobs):
        at = job * plan.job_bytes
        mine = words[job * plan.job_words:(job + 1) * plan.job_words]
        for k, word in enumerate(mine):
            host_offsets[job * (plan.job_words + 1) + k] = at
            host_bytes[at:at + len(word)] = word
            at += len(word)
        host_offsets[job * (plan.job_words + 1) + len(mine)] = at
    want = np.array([{b'a': 0, b'b': 1, b'c': 2, b'cc': 3, b'd': 4, b'e': 5}.get(w, MISSING) for w in words], dtype=np.uint32)
    total = plan.jobs * plan.job_bytes
    # word k's begin is entry(k), its end entry(k) + 1: poisoning entry(k) spoils words k - 1 (its end) and k (its begin)
    spoiled = set()
    for word, value in ((10, 0xFFFFFFFF), (70, total + 1), (200, 0xFFFFFFF0), (300, None), (500, 0x80000000), (839, total + 16)):
        if value is None:
            value = host_offsets[entry(word)] - 3   # runs backwards: in front of word 299's begin
        host_offsets[entry(word)] = value
        spoiled.update((word - 1, word))
    # (word 300 after its poisoning: begins three bytes early and ends where it should -- a longer string that is no key)
    assert library.memb_hip_words_commit(batch) == 0
    rows = torch.full((len(words),), 7, dtype=torch.int32, device='cuda')
    assert resolve(context, batch, rows.data_ptr(), None) == 0, library.memb_hip_last_error()
    torch.cuda.synchronize()
    got = rows.cpu().numpy().view(np.uint32)
    for word in range(len(words)):
        if word in spoiled:
            assert got[word] == MISSING, (word, got[word])
        else:
            assert got[word] == want[word], (word, got[word], want[word])
    # the context still works afterwards
    pack = library.memb_hip_words_pack
    pack.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    queries = [b'd', b'nope', b'a']
    array = (ctypes.c_char_p * len(queries))(*queries)
    assert pack(batch, array, None, len(queries)) == 0
    assert resolve(context, batch, rows.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert rows.cpu().numpy().view(np.uint32).tolist()[:3] == [4, MISSING, 0]
    # a failed begin leaves no batch behind: the range lookup refuses instead of launching on stale buffers
    assert begin(batch, 0x7FFFFFFF, 0, ctypes.byref(plan)) == 1
    in_range = library.memb_hip_resolve_range_device
    in_range.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
    assert in_range(context, batch, 0, 1, rows.data_ptr(), None) == 1
    assert resolve(context, batch, rows.data_ptr(), None) == 1
    library.memb_hip_words_destroy(batch)
    library.memb_hip_ctx_destroy(context)


def test_words_that_are_packed_already(native, make_model):
    """Reader.resolve_packed_device: one buffer of UTF-8 bytes + n + 1 offsets (a tokenizer's output) instead of a list of
    str -- from host memory (bytes, bytearray, numpy) through the pinned job regions, from device tensors in place; the
    answers of resolve_rows_device; several lookups of one batch on different streams before the next batch begins."""
    import torch
    path, words = make_model(30000, 300, 'trained', 4)
    reader = native.Reader(path)
    checker = oracle.OracleReader(path)
    rng = np.random.default_rng(13)
    for count in (0, 1, 63, 5000, 70000, 600000):   # (600 000: several runs of jobs, looked up while later ones are copied)
        queries = [words[i] for i in rng.integers(0, len(words), size=count)]
        for i in range(0, count, 9):
            queries[i] = queries[i] + 'x'
        if count > 100:
            queries[17] = ''
            queries[18] = 'naïve-日本語'
        encoded = [q.encode('utf-8') for q in queries]
        blob = b''.join(encoded)
        offsets = np.cumsum([0] + [len(e) for e in encoded]).astype(np.uint32)
        expected = checker.resolve_rows(queries) if count else np.zeros(0, dtype=np.uint32)
        for data in (blob, bytearray(blob), np.frombuffer(blob, dtype=np.uint8)):
            got = reader.resolve_packed_device(data, offsets)
            assert got.dtype == torch.int32 and got.device.type == 'cuda' and got.numel() == count
            assert np.array_equal(got.cpu().numpy().view(np.uint32), expected), count
        if count:
            on_device = reader.resolve_packed_device(torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).cuda(),
                                                     torch.from_numpy(offsets.view(np.int32).copy()).cuda())
            assert np.array_equal(on_device.cpu().numpy().view(np.uint32), expected), count
            assert np.array_equal(reader.resolve_rows_device(queries).cpu().numpy().view(np.uint32), expected)
    # one packed batch looked up on two streams, then the next batch at once: begin must wait for BOTH lookups
    queries = [words[i] for i in rng.integers(0, len(words), size=70000)]
    encoded = [q.encode('utf-8') for q in queries]
    blob = b''.join(encoded)
    offsets = np.cumsum([0] + [len(e) for e in encoded]).astype(np.uint32)
    expected = checker.resolve_rows(queries)
    first = reader.resolve_packed_device(blob, offsets)
    side = torch.cuda.Stream()
    second = torch.empty_like(first)
    with torch.cuda.stream(side):
        reader._impl.resolve_batch_to_device(reader._word_batch, second.data_ptr(), side.cuda_stream)
    third = reader.resolve_packed_device(b'zz', np.array([0, 2], dtype=np.uint32))
    torch.cuda.synchronize()
    assert np.array_equal(first.cpu().numpy().view(np.uint32), expected) and np.array_equal(second.cpu().numpy().view(np.uint32), expected)
    assert third.cpu().numpy().view(np.uint32).tolist() == [MISSING]
    with pytest.raises(ValueError):
        reader.resolve_packed_device(blob, np.array([0, 5, 3], dtype=np.uint32))          # runs backwards
    with pytest.raises(ValueError):
        reader.resolve_packed_device(b'abc', np.array([0, 2, 9], dtype=np.uint32))        # past the end of the bytes
    with pytest.raises(TypeError):
        reader.resolve_packed_device(blob, torch.zeros(3, dtype=torch.int32, device='cuda'))   # one on the host, one on the device
