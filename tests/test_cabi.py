"""The C-ABI shared library loads and exports what include/memb_hip.h declares.
No compute here: that needs the GPU (test_gpu_*.py)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import REPO


def declared_functions():
    text = open(os.path.join(REPO, 'include', 'memb_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(memb_hip_[a-z_]+)\s*\(', text)))


def test_header_declares_the_expected_entry_points():
    assert declared_functions() == sorted([
        'memb_hip_device_count', 'memb_hip_ctx_create_trained', 'memb_hip_ctx_create_uniform',
        'memb_hip_ctx_create_full', 'memb_hip_ctx_destroy', 'memb_hip_ctx_get_info', 'memb_hip_decode_rows',
        'memb_hip_decode_rows_device', 'memb_hip_decode_rows_device_ex', 'memb_hip_decode_rows_union_device', 'memb_hip_sync', 'memb_hip_algorithmic_bytes', 'memb_hip_last_error',
        'memb_hip_abi_version', 'memb_hip_ctx_set_option',
        'memb_hip_encoder_create', 'memb_hip_encoder_destroy', 'memb_hip_encoder_add_rows', 'memb_hip_encoder_counts',
        'memb_hip_encoder_pack', 'memb_hip_encoder_fetch', 'memb_hip_encoder_rows',
        # round 5: word -> row on the device, several batches in one launch
        'memb_hip_ctx_stage_words', 'memb_hip_words_create', 'memb_hip_words_destroy', 'memb_hip_words_pack',
        'memb_hip_words_begin', 'memb_hip_words_commit', 'memb_hip_words_count', 'memb_hip_resolve_rows_device',
        'memb_hip_resolve_range_device', 'memb_hip_resolve_range_union_device', 'memb_hip_resolve_packed_device',
        'memb_hip_decode_batches_device', 'memb_hip_decode_words',
    ])


def test_abi_version_and_options_without_compute(native):
    library = ctypes.CDLL(native.HIP_LIBRARY_PATH)
    header = open(os.path.join(REPO, 'include', 'memb_hip.h')).read()
    assert library.memb_hip_abi_version() == int(re.search(r'#define MEMB_HIP_ABI_VERSION (\d+)', header).group(1))
    library.memb_hip_last_error.restype = ctypes.c_char_p
    assert library.memb_hip_ctx_set_option(None, b'persistent', ctypes.c_uint64(1)) == 1   # MEMB_HIP_ERR_INVALID
    assert library.memb_hip_ctx_get_info(None, None) == 1


def test_header_is_plain_c_and_cxx():
    import subprocess
    header = os.path.join(REPO, 'include', 'memb_hip.h')
    for compiler, flags in (('gcc', ['-std=c99', '-pedantic', '-Wall', '-Werror', '-x', 'c']),
                            ('g++', ['-std=c++14', '-Wall', '-Werror', '-x', 'c++'])):
        result = subprocess.run([compiler, *flags, '-fsyntax-only', header], stdout=subprocess.PIPE,
                                stderr=subprocess.STDOUT, text=True)
        assert result.returncode == 0, result.stdout


def test_library_exports_every_declared_symbol(native):
    library = ctypes.CDLL(native.HIP_LIBRARY_PATH)
    for name in declared_functions():
        assert hasattr(library, name), name


def test_error_reporting_without_compute(native):
    library = ctypes.CDLL(native.HIP_LIBRARY_PATH)
    library.memb_hip_last_error.restype = ctypes.c_char_p
    count = ctypes.c_int(-1)
    code = library.memb_hip_device_count(ctypes.byref(count))
    assert (code == 0 and count.value > 0) or (code != 0 and count.value == 0)
    assert library.memb_hip_device_count(None) == 1  # MEMB_HIP_ERR_INVALID
    assert b'null' in library.memb_hip_last_error()
    context = ctypes.c_void_p()
    assert library.memb_hip_ctx_create_trained(ctypes.byref(context), 0, None) == 1
    assert not context.value
    assert library.memb_hip_sync(None) == 1
    library.memb_hip_ctx_destroy(None)  # harmless
    library.memb_hip_words_destroy(None)
    assert library.memb_hip_words_create(None, 0) == 1
    assert library.memb_hip_ctx_stage_words(None, None, ctypes.c_uint64(0), None, ctypes.c_uint64(0)) == 1
    assert library.memb_hip_resolve_rows_device(None, None, None, None) == 1
    assert library.memb_hip_words_begin(None, ctypes.c_size_t(1), ctypes.c_size_t(0), None) == 1
    assert library.memb_hip_words_commit(None) == 1
    assert library.memb_hip_resolve_range_device(None, None, ctypes.c_size_t(0), ctypes.c_size_t(0), None, None) == 1
    assert library.memb_hip_resolve_packed_device(None, None, None, ctypes.c_size_t(0), None, None) == 1
    assert library.memb_hip_decode_batches_device(None, None, ctypes.c_size_t(0), None) == 1
    assert library.memb_hip_decode_words(None, None, None, ctypes.c_size_t(0), ctypes.c_size_t(0)) == 1
    if count.value == 0:
        batch = ctypes.c_void_p()
        assert library.memb_hip_words_create(ctypes.byref(batch), 0) == 2 and not batch.value   # ERR_DEVICE
        assert b'no HIP device' in library.memb_hip_last_error()
    if count.value == 0:
        # a well-formed description still cannot be staged without a device
        import numpy as np

        class Desc(ctypes.Structure):
            _fields_ = [('dim', ctypes.c_uint32), ('n_rows', ctypes.c_uint64), ('packed_values', ctypes.c_void_p),
                        ('packed_values_bytes', ctypes.c_uint64), ('value_offsets', ctypes.c_void_p),
                        ('keys', ctypes.c_void_p), ('n_keys', ctypes.c_uint32), ('size_offsets', ctypes.c_void_p),
                        ('n_size_offsets', ctypes.c_uint32), ('centroids', ctypes.c_void_p),
                        ('n_centroids', ctypes.c_uint32), ('max_direct_bits', ctypes.c_uint32)]
        packed = np.array([0b01000000], dtype=np.uint8)
        offsets = np.zeros(1, dtype=np.uint32)
        keys = np.array([0, 1], dtype=np.uint8)
        size_offsets = np.array([0, 2], dtype=np.uint32)
        centroids = np.array([-1.0, 1.0], dtype=np.float32)
        desc = Desc(2, 1, packed.ctypes.data, 1, offsets.ctypes.data, keys.ctypes.data, 2,
                    size_offsets.ctypes.data, 2, centroids.ctypes.data, 2, 0)
        assert library.memb_hip_ctx_create_trained(ctypes.byref(context), 0, ctypes.byref(desc)) == 2  # ERR_DEVICE
        assert b'no HIP device' in library.memb_hip_last_error()


@pytest.mark.gpu
def test_pure_c_client(native, tmp_path):
    # a C99 program linked against the shared library: no Python, torch or C++ in the loop
    import subprocess
    library_dir = os.path.dirname(native.HIP_LIBRARY_PATH)
    binary = str(tmp_path / 'client')
    build = subprocess.run(
        ['gcc', '-std=c99', '-Wall', '-Werror', '-I', os.path.join(REPO, 'include'),
         os.path.join(REPO, 'tests', 'cabi', 'client.c'), '-L', library_dir, '-lmemb_hip',
         '-Wl,-rpath,' + library_dir, '-o', binary],
        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert build.returncode == 0, build.stdout
    run = subprocess.run([binary], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert run.returncode == 0, run.stdout
    lines = [line for line in run.stdout.splitlines() if line and not line.startswith('/opt/amdgpu')]
    assert lines[:4] == [
        '9 2.25 2.25 -1.5 -1.5 9',      # row 1 = 1100b
        '9 0 0 0 0 9',                  # missing row -> zeros, neighbours untouched
        '9 -1.5 2.25 -1.5 2.25 9',      # row 0 = 0101b
        '9 2.25 2.25 -1.5 -1.5 9',
    ], run.stdout
    assert lines[4] == 'error: ld must be at least col_off + dim'
    # round 5: five C strings through stage_words / words_pack / decode_words: beta, (unknown), alpha, (empty), beta
    assert [line for line in lines if line.startswith('words:')] == [
        'words: 2.25 2.25 -1.5 -1.5', 'words: 0 0 0 0', 'words: -1.5 2.25 -1.5 2.25', 'words: 0 0 0 0', 'words: 2.25 2.25 -1.5 -1.5'], run.stdout


def test_malformed_trained_descriptions_are_refused_before_any_device_work(native):
    # inconsistent decoder descriptions must come back as MEMB_HIP_ERR_INVALID with a message, on any
    # machine (the checks run before a device is opened); none may crash or hang
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

    def create(keys, size_offsets, n_centroids=255, offsets=(0,), dim=4):
        keys = np.array(keys, dtype=np.uint8)
        size_offsets = np.array(size_offsets, dtype=np.uint32)
        centroids = np.zeros(255, dtype=np.float32)
        offsets = np.array(offsets, dtype=np.uint32)
        packed = np.zeros(4, dtype=np.uint8)
        desc = Desc(dim, len(offsets), packed.ctypes.data, len(packed), offsets.ctypes.data, keys.ctypes.data, len(keys),
                    size_offsets.ctypes.data, len(size_offsets), centroids.ctypes.data, n_centroids, 0)
        context = ctypes.c_void_p()
        code = library.memb_hip_ctx_create_trained(ctypes.byref(context), 0, ctypes.byref(desc))
        if code == 0:
            library.memb_hip_ctx_destroy(context)
        return code, library.memb_hip_last_error().decode()

    INVALID = 1
    OVERSUBSCRIBED = (INVALID, 'Huffman code lengths describe no prefix code')
    assert create([1, 2, 3], [0, 3]) == OVERSUBSCRIBED                                              # three 1-bit codes
    # three 1-bit codes and a 13-bit one: the long code's first-level prefix lies past the table
    # (a heap overflow in buildDecodeTable before the Kraft check; the CPU ASan build runs this too)
    assert create([1, 2, 3, 4], [0, 3] + [3] * 11 + [4]) == OVERSUBSCRIBED
    assert create([1, 2, 3, 4, 5], [0, 1, 2, 5]) == OVERSUBSCRIBED                                  # 1 + 1 + 3 codes of 1, 2, 3 bits
    assert create([1, 2, 3], [0, 2, 1, 3])[0] == INVALID                                            # counts go down
    assert create(list(range(18)), list(range(18)) + [18])[0] == INVALID                            # a 17-bit code
    assert create([1, 200], [0, 2], n_centroids=10) == (INVALID, 'Huffman symbol without a centroid')
    assert create([1, 2], [0, 2], offsets=(100,)) == (INVALID, 'value offset beyond packed values')
    assert create([1, 2], [])[0] == INVALID
    assert create([1, 2], [0, 2], dim=0)[0] == INVALID
