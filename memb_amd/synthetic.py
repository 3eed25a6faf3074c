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
    if count == 0:
        return []
    # one row of characters per word -- letters, then the serial's digits (most significant first,
    # no leading zeros, as numpy.base_repr(i, 36).lower() prints them), then spaces -- split at the end
    serial = np.arange(count, dtype=np.int64)
    digit_count = np.ones(count, dtype=np.int64)
    power = 36
    while power <= count - 1:
        digit_count += serial >= power
        power *= 36
    width = 12 + int(digit_count.max()) + 1
    rows = np.full((count, width), ord(' '), dtype=np.uint8)
    columns = np.arange(12)
    mask = columns[None, :] < lengths[:, None]
    rows[:, :12][mask] = letters[mask]
    alphabet = np.frombuffer(b'0123456789abcdefghijklmnopqrstuvwxyz', dtype=np.uint8)
    index = np.arange(count)
    for position in range(int(digit_count.max())):           # position counted from the least significant digit
        has = digit_count > position
        digit = (serial[has] // 36 ** position) % 36
        rows[index[has], lengths[has] + digit_count[has] - 1 - position] = alphabet[digit]
    return rows.tobytes().decode('ascii').split()


def make_vectors(count, dim, seed=1234, distribution='normal'):
    '''one block of vectors from one generator (small cases; build_file draws large models in blocks)'''
    rng = np.random.default_rng(seed)
    if distribution == 'normal':
        return (rng.standard_normal((count, dim), dtype=np.float32) * np.float32(0.4))
    if distribution == 'student':
        return (rng.standard_t(5, size=(count, dim)) * 0.3).astype(np.float32)
    raise ValueError('unknown distribution ' + distribution)


def _vector_block(seed, block, rows, dim, distribution):
    # block 0 continues the stream build_file always used (small files, the committed fixtures);
    # later blocks have generators of their own so that they can be drawn on other threads
    rng = np.random.default_rng(seed if block == 0 else [seed, block])
    if distribution == 'normal':
        return rng.standard_normal((rows, dim), dtype=np.float32) * np.float32(0.4)
    return (rng.standard_t(5, size=(rows, dim)) * 0.3).astype(np.float32)


def build_file(path, count, dim=300, storage_type='trained', bits_per_weight=4,
               seed=1234, word_seed=7, distribution='normal', slice_words=200000, device=None):
    '''Write a synthetic model; returns the words in insertion order.
    Vectors are drawn in blocks of `slice_words` words on a few threads (numpy's generators release
    the GIL) while the main thread hands finished blocks to the builder in order.
    device: HIP device for the builder's bulk work (memb_amd.Builder); the file is the same either way.'''
    from concurrent.futures import ThreadPoolExecutor
    if distribution not in ('normal', 'student'):
        raise ValueError('unknown distribution ' + distribution)
    words = make_words(count, word_seed)
    builder = Builder(dim, storage_type, bits_per_weight, device=device)
    starts = list(range(0, count, slice_words))
    workers = max(1, min(16, os.cpu_count() or 1, len(starts)))
    with ThreadPoolExecutor(max_workers=workers) as pool:
        pending = []
        submitted = 0
        for position, start in enumerate(starts):
            while submitted < len(starts) and submitted < position + workers + 1:   # bounded look-ahead
                rows = min(count, starts[submitted] + slice_words) - starts[submitted]
                pending.append(pool.submit(_vector_block, seed, submitted, rows, dim, distribution))
                submitted += 1
            block = pending[position].result()
            pending[position] = None
            builder.add_words(words[start:start + len(block)], block)
    tmp = str(path) + '.tmp{}'.format(os.getpid())
    builder.save(tmp)
    os.replace(tmp, str(path))
    return words


def cache_dir():
    return os.environ.get('MEMB_BENCH_CACHE', '/tmp/memb_amd_bench')


def cached_model_path(count, dim=300, storage_type='trained', bits_per_weight=4, seed=1234, distribution='normal'):
    # g2: generator version -- build_file draws blocks >= 1 from generators of their own since round 2, so a
    # model of more than 200 000 words written by an older tree has other contents under the old name
    name = 'synthetic_g2_{}w_{}d_{}{}bit_{}_{}.bin'.format(count, dim, storage_type, bits_per_weight, distribution, seed)
    return os.path.join(cache_dir(), name)


def cached_model(count, dim=300, storage_type='trained', bits_per_weight=4, seed=1234, distribution='normal', device=None):
    '''Path of a synthetic model in the per-box cache, written on first use (device: see build_file;
    MEMB_SYNTH_DEVICE in the environment supplies a default). Returns (path, seconds spent building;
    0.0 when it was already there)'''
    import time
    path = cached_model_path(count, dim, storage_type, bits_per_weight, seed, distribution)
    if os.path.exists(path):
        return path, 0.0
    os.makedirs(cache_dir(), exist_ok=True)
    if device is None and os.environ.get('MEMB_SYNTH_DEVICE', '') != '':
        device = int(os.environ['MEMB_SYNTH_DEVICE'])
    start = time.time()
    build_file(path, count, dim, storage_type, bits_per_weight, seed=seed, distribution=distribution, device=device)
    return path, time.time() - start
