"""Write side: collect (word, vector) pairs and write one memb file.

Same constructor and methods as the reference's `memb.Builder`
(python/memb/builder.py:4-39). Here the writer exists to produce the files the
lookup path reads -- test fixtures and benchmark models -- and follows the
reference's encoders step by step (memb_amd/csrc/compression_strategy.cpp).
"""
import os

from . import _memb


class Builder:
    """Accumulates vectors and writes them, compressed, on `save`.

    dim              length every vector must have
    storage_type     'trained' (k-means codebook + Huffman), 'uniform'
                     (per-word min/max, one byte per weight) or 'full' (fp32)
    bits_per_weight  precision asked of the storage; values a storage cannot
                     honour are clamped by it
    device           (not in the reference API) HIP device index: the trained storage then quantises,
                     counts and bit-packs on that GPU -- vectors stream to it as they are added, only
                     the k-means fit on the first 10 000 words and the Huffman tree stay on the host.
                     The file is byte for byte the one the host writes. None = host, as the reference.
    """

    def __init__(self, dim, storage_type='trained', bits_per_weight=4, device=None):
        if device is None:
            self._native = _memb.Builder(dim, storage_type, bits_per_weight)
        else:
            self._native = _memb.Builder(dim, storage_type, bits_per_weight, int(device))

    def add_word(self, word, vector):
        """One word; `vector` is a float32 sequence of length dim. Raises on a
        wrong length or a word added before."""
        self._native.add_word(word, vector)

    def add_words(self, words, matrix):
        """Many words in one call: `matrix` is float32 of shape (len(words), dim).
        Not part of the reference API; same checks as add_word, row by row."""
        self._native.add_words(words, matrix)

    def save(self, filename):
        """Compress what was added and write the file (str or path-like)."""
        self._native.save(os.fspath(filename))
