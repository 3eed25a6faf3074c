"""ctypes binding of the CPU checker (oracle/memb_oracle.c, oracle/_ref).

TEST INFRASTRUCTURE ONLY. May be imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg -- never by the
memb_amd package.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_LIBRARY = os.environ.get('MEMB_ORACLE_LIBRARY', os.path.join(_HERE, 'libmemb_oracle.so'))  # override: sanitizer build
REFERENCE_LIBRARY = os.path.join(_HERE, '_ref', 'libmemb_ref.so')
UNIFORM_EXPR_LIBRARY = os.path.join(_HERE, 'libmemb_uniform_expr.so')

_u8p = ctypes.POINTER(ctypes.c_uint8)
_u16p = ctypes.POINTER(ctypes.c_uint16)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_f32p = ctypes.POINTER(ctypes.c_float)


def _ptr(array, kind):
    return array.ctypes.data_as(kind)


def build():
    """Compile the checker with the committed recipe (oracle/Makefile)."""
    subprocess.run(['make', '-s', '-C', _HERE], check=True)


def _load_oracle():
    if not os.path.exists(ORACLE_LIBRARY):
        build()
    lib = ctypes.CDLL(ORACLE_LIBRARY)
    lib.memb_oracle_open.restype = ctypes.c_void_p
    lib.memb_oracle_open.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t]
    lib.memb_oracle_close.argtypes = [ctypes.c_void_p]
    lib.memb_oracle_dim.restype = ctypes.c_uint32
    lib.memb_oracle_dim.argtypes = [ctypes.c_void_p]
    lib.memb_oracle_storage_type.restype = ctypes.c_uint32
    lib.memb_oracle_storage_type.argtypes = [ctypes.c_void_p]
    lib.memb_oracle_size.restype = ctypes.c_size_t
    lib.memb_oracle_size.argtypes = [ctypes.c_void_p]
    lib.memb_oracle_key.restype = ctypes.c_char_p
    lib.memb_oracle_key.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    lib.memb_oracle_resolve.restype = ctypes.c_int
    lib.memb_oracle_resolve.argtypes = [ctypes.c_void_p, ctypes.c_char_p, _u32p]
    lib.memb_oracle_resolve_many.restype = None
    lib.memb_oracle_resolve_many.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_char_p), ctypes.c_size_t, _u32p]
    lib.memb_oracle_word_embedding.argtypes = [ctypes.c_void_p, ctypes.c_char_p, _f32p]
    lib.memb_oracle_batch_embedding.argtypes = [
        ctypes.c_void_p, ctypes.POINTER(ctypes.c_char_p), ctypes.c_size_t, _f32p]
    lib.memb_oracle_rows_embedding.argtypes = [
        ctypes.c_void_p, _u32p, ctypes.c_size_t, _f32p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t]
    lib.memb_oracle_stream_bytes.restype = ctypes.c_uint32
    lib.memb_oracle_stream_bytes.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
    lib.memb_oracle_uniform_value.restype = ctypes.c_float
    lib.memb_oracle_uniform_value.argtypes = [ctypes.c_float, ctypes.c_float, ctypes.c_uint8, ctypes.c_uint8]
    _declare_codec(lib, 'memb_oracle')
    lib.memb_oracle_decoder_sizes.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
    return lib


def _declare_codec(lib, prefix):
    create = getattr(lib, prefix + '_decoder_create')
    create.restype = ctypes.c_void_p
    create.argtypes = [_u8p, ctypes.c_size_t, _u32p, ctypes.c_size_t, ctypes.c_uint32]
    getattr(lib, prefix + '_decoder_destroy').argtypes = [ctypes.c_void_p]
    decode = getattr(lib, prefix + '_decode_symbols')
    decode.argtypes = [ctypes.c_void_p, _u8p, ctypes.c_size_t, ctypes.c_size_t, _u8p]
    codes = getattr(lib, prefix + '_canonical_codes')
    codes.argtypes = [_u8p, _u32p, ctypes.c_size_t, _u16p, _u32p]
    pack = getattr(lib, prefix + '_bitstream_pack')
    pack.restype = ctypes.c_size_t
    pack.argtypes = [_u16p, _u32p, ctypes.c_size_t, _u8p, ctypes.c_size_t]


_oracle = None
_reference = None


def oracle_library():
    global _oracle
    if _oracle is None:
        _oracle = _load_oracle()
    return _oracle


def reference_available():
    return os.path.exists(REFERENCE_LIBRARY)


def reference_library():
    """The reference's own decode headers (oracle/_ref); None when not built."""
    global _reference
    if _reference is None and reference_available():
        _reference = ctypes.CDLL(REFERENCE_LIBRARY)
        _declare_codec(_reference, 'memb_ref')
    return _reference


class Codec:
    """Table decoder + canonical codes + bit packer of either implementation."""

    def __init__(self, which='oracle'):
        if which == 'oracle':
            self.lib, self.prefix = oracle_library(), 'memb_oracle'
        else:
            self.lib, self.prefix = reference_library(), 'memb_ref'
            if self.lib is None:
                raise RuntimeError('oracle/_ref is not built')

    def _fn(self, name):
        return getattr(self.lib, self.prefix + '_' + name)

    def canonical_codes(self, keys, lengths):
        keys = np.ascontiguousarray(keys, dtype=np.uint8)
        lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
        codes = np.zeros(256, dtype=np.uint16)
        bits = np.zeros(256, dtype=np.uint32)
        self._fn('canonical_codes')(_ptr(keys, _u8p), _ptr(lengths, _u32p), len(keys), _ptr(codes, _u16p), _ptr(bits, _u32p))
        return codes, bits

    def bitstream_pack(self, codes, bits):
        codes = np.ascontiguousarray(codes, dtype=np.uint16)
        bits = np.ascontiguousarray(bits, dtype=np.uint32)
        out = np.zeros(2 * len(codes) + 8, dtype=np.uint8)
        size = self._fn('bitstream_pack')(_ptr(codes, _u16p), _ptr(bits, _u32p), len(codes), _ptr(out, _u8p), len(out))
        assert size != ctypes.c_size_t(-1).value
        return out[:size].copy()

    def decode_symbols(self, keys, size_offsets, max_direct_bits, stream, count):
        keys = np.ascontiguousarray(keys, dtype=np.uint8)
        size_offsets = np.ascontiguousarray(size_offsets, dtype=np.uint32)
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        padded = np.concatenate([stream, np.zeros(1, dtype=np.uint8)])  # valid pointer for empty streams
        decoder = self._fn('decoder_create')(
            _ptr(keys, _u8p), len(keys), _ptr(size_offsets, _u32p), len(size_offsets), max_direct_bits)
        try:
            out = np.zeros(count, dtype=np.uint8)
            self._fn('decode_symbols')(decoder, _ptr(padded, _u8p), len(stream), count, _ptr(out, _u8p))
        finally:
            self._fn('decoder_destroy')(decoder)
        return out


class OracleReader:
    """The restated memb::Reader (CPU)."""

    def __init__(self, filename, num_threads=0, max_direct_bits=0):
        self.lib = oracle_library()
        error = ctypes.create_string_buffer(256)
        self.handle = self.lib.memb_oracle_open(str(filename).encode(), num_threads, max_direct_bits, error, 256)
        if not self.handle:
            raise RuntimeError(error.value.decode())

    def close(self):
        if self.handle:
            self.lib.memb_oracle_close(self.handle)
            self.handle = None

    def __del__(self):
        self.close()

    @property
    def dim(self):
        return self.lib.memb_oracle_dim(self.handle)

    @property
    def storage_type(self):
        return self.lib.memb_oracle_storage_type(self.handle)

    def __len__(self):
        return self.lib.memb_oracle_size(self.handle)

    def keys(self):
        return [self.lib.memb_oracle_key(self.handle, i).decode() for i in range(len(self))]

    def resolve(self, word):
        row = ctypes.c_uint32(0)
        found = self.lib.memb_oracle_resolve(self.handle, word.encode(), ctypes.byref(row))
        return row.value if found else None

    def resolve_rows(self, words):
        """the reference's search, word by word (memb_oracle_resolve); 0xFFFFFFFF = not in the model. A str is
        searched as its UTF-8 bytes up to the first NUL, which is where strcmp stops"""
        rows = np.empty(len(words), dtype=np.uint32)
        array = (ctypes.c_char_p * max(len(words), 1))(*[w.encode() if isinstance(w, str) else w for w in words])
        self.lib.memb_oracle_resolve_many(self.handle, array, len(words), _ptr(rows, _u32p))
        return rows

    def word_embedding(self, word):
        out = np.empty(self.dim, dtype=np.float32)
        self.lib.memb_oracle_word_embedding(self.handle, word.encode(), _ptr(out, _f32p))
        return out

    def batch_embedding(self, words):
        out = np.empty((len(words), self.dim), dtype=np.float32)
        array = (ctypes.c_char_p * max(len(words), 1))(*[w.encode() for w in words])
        self.lib.memb_oracle_batch_embedding(self.handle, array, len(words), _ptr(out, _f32p))
        return out

    def rows_embedding(self, rows, out=None, col_off=0, num_threads=0):
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        if out is None:
            out = np.empty((len(rows), self.dim), dtype=np.float32)
        assert out.dtype == np.float32 and out.flags.c_contiguous and out.shape[0] == len(rows)
        self.lib.memb_oracle_rows_embedding(
            self.handle, _ptr(rows, _u32p), len(rows), _ptr(out, _f32p), out.shape[1], col_off, num_threads)
        return out

    def stream_bytes(self, row):
        return self.lib.memb_oracle_stream_bytes(self.handle, int(row))

    def trained_view(self):
        """Pointers to the trained storage's arrays inside the mapped file (None for other storages)."""
        view = TrainedArrays()
        self.lib.memb_oracle_trained_view.argtypes = [ctypes.c_void_p, ctypes.POINTER(TrainedArrays)]
        self.lib.memb_oracle_trained_view.restype = ctypes.c_int
        return view if self.lib.memb_oracle_trained_view(self.handle, ctypes.byref(view)) else None


class TrainedArrays(ctypes.Structure):
    _fields_ = [('packed_values', ctypes.c_void_p), ('packed_values_size', ctypes.c_uint64),
                ('value_offsets', ctypes.c_void_p), ('word_count', ctypes.c_uint64),
                ('keys', ctypes.c_void_p), ('key_count', ctypes.c_uint64),
                ('size_offsets', ctypes.c_void_p), ('size_offset_count', ctypes.c_uint64),
                ('centroids', ctypes.c_void_p), ('centroid_count', ctypes.c_uint64)]


class ReferenceDecoder:
    """Rows -> fp32 with the REFERENCE's HuffmanTableDecoder (oracle/_ref) over the
    arrays of a file opened by OracleReader: the decode loop of
    TrainedCompressedStorage::extract (reference src/trained_compression.cpp:129-137)."""

    DEFAULT_DECODE_TABLE_BIT_LENGTH = 10   # reference src/trained_compression.h:11

    def __init__(self, reader, max_direct_bits=DEFAULT_DECODE_TABLE_BIT_LENGTH):
        self.lib = reference_library()
        if self.lib is None:
            raise RuntimeError('oracle/_ref is not built')
        self.reader = reader   # keeps the mapping alive
        self.view = reader.trained_view()
        if self.view is None:
            raise RuntimeError('not a trained storage')
        self.dim = reader.dim
        self.lib.memb_ref_rows_embedding.restype = None
        self.lib.memb_ref_rows_embedding.argtypes = [
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
            ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32]
        self.decoder = self.lib.memb_ref_decoder_create(
            ctypes.cast(self.view.keys, _u8p), self.view.key_count,
            ctypes.cast(self.view.size_offsets, _u32p), self.view.size_offset_count, max_direct_bits)

    def close(self):
        if self.decoder:
            self.lib.memb_ref_decoder_destroy(self.decoder)
            self.decoder = None

    def __del__(self):
        self.close()

    def rows_embedding(self, rows, out=None, num_threads=1):
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        if out is None:
            out = np.empty((len(rows), self.dim), dtype=np.float32)
        assert out.dtype == np.float32 and out.flags.c_contiguous and out.shape[0] == len(rows)
        view = self.view
        self.lib.memb_ref_rows_embedding(
            self.decoder, view.packed_values, view.packed_values_size, view.value_offsets, view.word_count,
            view.centroids, rows.ctypes.data, len(rows), self.dim, out.ctypes.data, out.shape[1], num_threads)
        return out


_uniform_expr_library = None


def uniform_expression(min_value, max_value, levels, values):
    """oracle/uniform_expr.cpp: the reference's expression over a row of bytes, as compiled with the
    reference's flags (g++ -std=c++14 -O3, baseline x86-64). float32 array of len(values)."""
    global _uniform_expr_library
    if _uniform_expr_library is None:
        if not os.path.exists(UNIFORM_EXPR_LIBRARY):
            build()
        _uniform_expr_library = ctypes.CDLL(UNIFORM_EXPR_LIBRARY)
        _uniform_expr_library.memb_uniform_expr.restype = None
        _uniform_expr_library.memb_uniform_expr.argtypes = [
            ctypes.c_float, ctypes.c_float, ctypes.c_uint8, _u8p, ctypes.c_size_t, _f32p]
    values = np.ascontiguousarray(values, dtype=np.uint8)
    out = np.empty(len(values), dtype=np.float32)
    _uniform_expr_library.memb_uniform_expr(
        ctypes.c_float(min_value), ctypes.c_float(max_value), int(levels), _ptr(values, _u8p), len(values), _ptr(out, _f32p))
    return out


def uniform_value(min_value, max_value, value, levels):
    return oracle_library().memb_oracle_uniform_value(min_value, max_value, value, levels)
