"""Synthetic models with the shapes of the public embedding files.

Real GloVe / fastText files are not available offline; benchmarks and tests
use files written by memb_amd.Builder from seeded random data (SURVEY.md
section 8d): unique lower-case words, vectors i.i.d. N(0, 0.4^2) or a
heavier-tailed Student-t(5) * 0.3.
"""
import os

import numpy as np

from .builder import Builder


def make_words(count, seed=7):
    '''`count` unique words: 3-12 random lower-case letters plus a base-36 serial'''
    rng = np.random.default_rng(seed)
    lengths = rng.integers(3, 13, size=count)
    letters = rng.integers(0, 26, size=(count, 12)).astype(np.uint8) + ord('a')
    words = []
    for i in range(count):
        words.append(letters[i, :lengths[i]].tobytes().decode('ascii') + np.base_repr(i, 36).lower())
    return words


def make_vectors(count, dim, seed=1234, distribution='normal'):
    rng = np.random.default_rng(seed)
    if distribution == 'normal':
        return (rng.standard_normal((count, dim), dtype=np.float32) * np.float32(0.4))
    if distribution == 'student':
        return (rng.standard_t(5, size=(count, dim)) * 0.3).astype(np.float32)
    raise ValueError('unknown distribution ' + distribution)


def build_file(path, count, dim=300, storage_type='trained', bits_per_weight=4,
               seed=1234, word_seed=7, distribution='normal', slice_words=200000):
    '''Write a synthetic model; returns the words in insertion order'''
    words = make_words(count, word_seed)
    builder = Builder(dim, storage_type, bits_per_weight)
    rng = np.random.default_rng(seed)
    for start in range(0, count, slice_words):
        stop = min(count, start + slice_words)
        if distribution == 'normal':
            block = rng.standard_normal((stop - start, dim), dtype=np.float32) * np.float32(0.4)
        else:
            block = (rng.standard_t(5, size=(stop - start, dim)) * 0.3).astype(np.float32)
        builder.add_words(words[start:stop], block)
    tmp = str(path) + '.tmp{}'.format(os.getpid())
    builder.save(tmp)
    os.replace(tmp, str(path))
    return words


def cache_dir():
    return os.environ.get('MEMB_BENCH_CACHE', '/tmp/memb_amd_bench')


def cached_model(count, dim=300, storage_type='trained', bits_per_weight=4, seed=1234, distribution='normal'):
    '''Path of a synthetic model in the per-box cache, written on first use.
    Returns (path, seconds spent building; 0.0 when it was already there)'''
    import time
    name = 'synthetic_{}w_{}d_{}{}bit_{}_{}.bin'.format(count, dim, storage_type, bits_per_weight, distribution, seed)
    path = os.path.join(cache_dir(), name)
    if os.path.exists(path):
        return path, 0.0
    os.makedirs(cache_dir(), exist_ok=True)
    start = time.time()
    build_file(path, count, dim, storage_type, bits_per_weight, seed=seed, distribution=distribution)
    return path, time.time() - start
