"""The product's host decode path (memb_amd/csrc/compression_strategy.cpp: extractRowHost,
decodeRowsHost) -- the reference's serial / threaded CPU reader (src/reader.cpp:49-86) restated
inside the product for hosts without a GPU (BASELINE.json configs[0]) and, when asked for with
`host_below`, for single words. It is only ever taken on request: Reader(..., device='cpu'),
MEMB_HIP_DEVICE=cpu or a non-zero host_below. Checked here, without a GPU, against the CPU checker
and the rows the reference's own decoder produced (tests/golden/*.rows.npy)."""
import os
import subprocess

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, REPO, bits_equal, golden_json


def host_reader(native, path, **arguments):
    reader = native.Reader(path, device='cpu', **arguments)
    assert reader.device == 'cpu'
    return reader


@pytest.mark.parametrize('max_direct_bits', [0, 1, 3, 12])
def test_golden_trained_models_on_the_host(native, max_direct_bits):
    # rows decoded by the reference's HuffmanTableDecoder (oracle/_ref, tests/golden/make_golden.py)
    for entry in golden_json('models.json'):
        if entry['storage'] != 'trained':
            continue
        path = os.path.join(GOLDEN, entry['file'])
        reader = host_reader(native, path, max_direct_decode_bits=max_direct_bits)
        assert reader.keys() == entry['keys']
        expected = np.load(os.path.join(GOLDEN, entry['rows']))
        assert bits_equal(reader[entry['keys']], expected)
        assert bits_equal(reader[entry['keys'][-1]], expected[-1])
        assert reader.host_rows_decoded == len(entry['keys']) + 1


@pytest.mark.parametrize('storage,bits', [('trained', 2), ('trained', 4), ('trained', 6), ('trained', 8), ('uniform', 1),
                                          ('uniform', 8), ('full', 8)])
def test_host_batches_equal_the_checker(native, make_model, storage, bits):
    path, words = make_model(3000, 300, storage, bits, distribution='student' if bits == 8 else 'normal')
    checker = oracle.OracleReader(path)
    reader = host_reader(native, path)
    rng = np.random.default_rng(3)
    keys = sorted(words)
    for count in (0, 1, 2, 63, 1023, 1024, 1025, 2600):
        batch = [keys[i] if rng.random() > 0.1 else 'missing-%d' % i for i in rng.integers(0, len(keys), size=count)]
        got = reader.batch_embedding(batch)
        assert got.shape == (count, 300) and got.dtype == np.float32
        assert bits_equal(got, checker.batch_embedding(batch)), count
    assert bits_equal(reader['not a word'], np.zeros(300, dtype=np.float32))
    # strided output: other columns stay as they were
    batch = keys[:1500]
    wide = np.full((len(batch), 610), 7.5, dtype=np.float32)
    reader.batch_embedding_into(batch, wide, 305)
    assert bits_equal(wide[:, 305:605], checker.batch_embedding(batch))
    assert (wide[:, :305] == 7.5).all() and (wide[:, 605:] == 7.5).all()


def test_host_threads_do_not_change_results(native, make_model):
    # reference src/tests.cpp:90-113: a 1025-word batch, one thread against four, bit for bit
    path, words = make_model(3000, 300, 'trained', 4)
    batch = [sorted(words)[(7 * i) % 3000] for i in range(1025)]
    serial = native.Reader(path, 1, device='cpu').batch_embedding(batch)
    for threads in (2, 4, 0):
        assert bits_equal(native.Reader(path, threads, device='cpu').batch_embedding(batch), serial)
    assert bits_equal(serial, oracle.OracleReader(path, 1).batch_embedding(batch))


def test_uniform_rows_with_odd_ranges_on_the_host(native, tmp_path):
    # per-word ranges that stress the four fp32 operations: subnormal spans, huge spans, constant rows
    # (max == min: 0 / 0 at encode, mirrored), negative-only rows; host path and checker agree bit for bit
    rng = np.random.default_rng(9)
    builder = native.Builder(16, 'uniform', 8)
    rows = {
        'subnormal': (rng.random(16) * 3e-39).astype(np.float32),
        'tiny': (rng.standard_normal(16) * 1e-30).astype(np.float32),
        'huge': (rng.standard_normal(16) * 1e38).astype(np.float32),
        'constant': np.full(16, 0.25, dtype=np.float32),
        'negative': (-rng.random(16) - 1).astype(np.float32),
        'mixed': rng.standard_normal(16).astype(np.float32),
    }
    with np.errstate(all='ignore'):
        for word, vector in rows.items():
            builder.add_word(word, vector)
    path = str(tmp_path / 'odd_uniform.bin')
    builder.save(path)
    batch = sorted(rows) + ['nope']
    with np.errstate(all='ignore'):
        got = host_reader(native, path)[batch]
        want = oracle.OracleReader(path).batch_embedding(batch)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert bits_equal(np.nan_to_num(got, nan=1.0), np.nan_to_num(want, nan=1.0))


def test_device_methods_refuse_a_host_reader(native):
    reader = host_reader(native, os.path.join(GOLDEN, 'synthetic_4bit.bin'))
    assert reader.info()['device'] == 'cpu'
    with pytest.raises(RuntimeError, match='decodes on the host'):
        reader._impl.context_handle()


def test_host_below_keeps_small_host_batches_on_the_host(native, monkeypatch):
    # asked for explicitly, small host batches never reach the device -- so they also work where there is none
    path = os.path.join(GOLDEN, 'synthetic_4bit.bin')
    checker = oracle.OracleReader(path)
    reader = native.Reader(path, device=0, host_below=8)
    keys = reader.keys()
    assert reader.device == 0 and reader.host_rows_decoded == 0
    assert bits_equal(reader[keys[3]], checker.word_embedding(keys[3]))
    assert bits_equal(reader[keys[:8]], checker.batch_embedding(keys[:8]))
    assert reader.host_rows_decoded == 9
    if native.hip_device_count() == 0:
        with pytest.raises(RuntimeError, match='no HIP device available'):   # 9 words: the device's, and there is none
            reader[keys[:9]]
        with pytest.raises(RuntimeError, match='no HIP device available'):   # 5000 words: searched on the device too
            reader[[keys[i % len(keys)] for i in range(5000)]]
    # the device word search and device buffers are not for readers that decode on the host, and say so
    host = native.Reader(path, device='cpu')
    with pytest.raises(RuntimeError, match="decodes on the host"):
        host.resolve_rows_device(keys[:3])
    with pytest.raises(RuntimeError, match="decodes on the host"):
        host.rows_embedding_device_many([])
    with pytest.raises(RuntimeError, match="decodes on the host"):
        host.stage_words()
    assert bits_equal(host[[keys[i % len(keys)] for i in range(5000)]], checker.batch_embedding([keys[i % len(keys)] for i in range(5000)]))
    monkeypatch.setenv('MEMB_HOST_BELOW', '5')
    assert native.Reader(path)._impl.host_below() == 5
    monkeypatch.setenv('MEMB_HIP_DEVICE', 'cpu')
    assert native.Reader(path).device == 'cpu'


def build_reader_tests(native, tmp_path, extra_flags):
    binary = str(tmp_path / 'reader_tests')
    library_dir = os.path.dirname(native.HIP_LIBRARY_PATH)
    command = ['g++', '-std=c++17', '-Wall', '-Werror', '-ffp-contract=off'] + extra_flags
    command += ['-I', os.path.join(REPO, 'include'), os.path.join(REPO, 'tests', 'cpp', 'reader_tests.cpp')]
    command += [os.path.join(REPO, 'memb_amd', 'csrc', name) for name in ('reader.cpp', 'builder.cpp', 'compression_strategy.cpp')]
    command += ['-L', library_dir, '-lmemb_hip', '-Wl,-rpath,' + library_dir, '-pthread', '-o', binary]
    build = subprocess.run(command, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert build.returncode == 0, build.stdout
    return binary


def test_cpp_cases_under_address_sanitizer(native, tmp_path):
    # wire parser, table construction (incl. descriptions of impossible codes), writer and host decode
    # of the product, instrumented: AddressSanitizer + UBSan, CPU only (the GPU pool has no sanitizer runs)
    binary = build_reader_tests(native, tmp_path, ['-O1', '-g', '-fsanitize=address,undefined',
                                                   '-fno-sanitize-recover=undefined', '-fno-omit-frame-pointer'])
    env = dict(os.environ, MEMB_HIP_DEVICE='cpu', ASAN_OPTIONS='detect_leaks=0:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1')
    run = subprocess.run([binary], cwd=str(tmp_path), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                         timeout=600)
    assert run.returncode == 0, run.stdout[-4000:]
    assert 'malformed code descriptions' in run.stdout and run.stdout.strip().endswith('ok (0 failed checks)')
    assert 'runtime error' not in run.stdout and 'AddressSanitizer' not in run.stdout


def test_reference_cpp_cases_on_the_host_path(native, tmp_path):
    # the reference's own test cases (tests/cpp/reader_tests.cpp restates src/tests.cpp) with the
    # host path selected through the environment: six words x three storages, the forced two-level
    # table, 1025 words serial vs threaded, the refusals
    binary = str(tmp_path / 'reader_tests')
    library_dir = os.path.dirname(native.HIP_LIBRARY_PATH)
    command = ['g++', '-O2', '-std=c++17', '-Wall', '-Werror', '-ffp-contract=off', '-I', os.path.join(REPO, 'include'),
               os.path.join(REPO, 'tests', 'cpp', 'reader_tests.cpp')]
    command += [os.path.join(REPO, 'memb_amd', 'csrc', name) for name in ('reader.cpp', 'builder.cpp', 'compression_strategy.cpp')]
    command += ['-L', library_dir, '-lmemb_hip', '-Wl,-rpath,' + library_dir, '-pthread', '-o', binary]
    build = subprocess.run(command, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert build.returncode == 0, build.stdout
    run = subprocess.run([binary], cwd=str(tmp_path), env=dict(os.environ, MEMB_HIP_DEVICE='cpu'),
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert run.returncode == 0, run.stdout
    assert 'trained storage, first-level table of 1 bit' in run.stdout and run.stdout.strip().endswith('ok (0 failed checks)')


def test_to_keyed_vectors_decodes_rows_by_number(native, monkeypatch):
    """BaseReader.to_keyed_vectors (reference python/memb/reader.py:19-30) with a stand-in gensim (not installed here):
    Reader decodes rows 0 .. N-1 instead of looking every key up again -- the same vocabulary and matrix as the
    reference's batch_embedding(keys()); a ReadersUnion goes the reference's way."""
    import sys
    import types

    class KeyedVectors:
        def __init__(self, vector_size):
            self.vector_size = vector_size

        def add(self, keys, vectors):
            self.keys, self.vectors = list(keys), np.asarray(vectors)

    gensim = types.ModuleType('gensim')
    gensim.models = types.ModuleType('gensim.models')
    gensim.models.KeyedVectors = KeyedVectors
    monkeypatch.setitem(sys.modules, 'gensim', gensim)
    monkeypatch.setitem(sys.modules, 'gensim.models', gensim.models)
    path = os.path.join(GOLDEN, 'synthetic_4bit.bin')
    reader = native.Reader(path, device='cpu')
    checker = oracle.OracleReader(path)
    exported = reader.to_keyed_vectors()
    assert exported.vector_size == reader.dim and exported.keys == checker.keys()
    assert bits_equal(exported.vectors, checker.batch_embedding(checker.keys()))
    union = native.ReadersUnion([reader, reader], 'concatenate').to_keyed_vectors()
    assert union.vector_size == 2 * reader.dim and bits_equal(union.vectors[:, :reader.dim], exported.vectors)
    monkeypatch.delitem(sys.modules, 'gensim')
    monkeypatch.delitem(sys.modules, 'gensim.models')
    monkeypatch.setitem(sys.modules, 'gensim', None)   # import gensim -> ImportError
    with pytest.raises(ImportError, match='install gensim'):
        reader.to_keyed_vectors()
