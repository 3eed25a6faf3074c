"""BASELINE.json's full-size configuration (GloVe-840B shape: 2,196,017 words x
300, trained 4-bit) on the GPU: the whole dump is compared with the CPU checker
slice by slice, plus size-independent properties (every value is a centroid,
permutation consistency, idempotence)."""
import os

import numpy as np
import pytest

import oracle
from conftest import bits_equal

pytestmark = pytest.mark.gpu

FULL_VOCAB = 2196017


@pytest.fixture(scope='module')
def full_model(native, tmp_path_factory):
    from memb_amd import synthetic
    count = int(os.environ.get('MEMB_TEST_FULL_VOCAB', FULL_VOCAB))
    path, _ = synthetic.cached_model(count, 300, 'trained', 4)   # shared with bench.py on the same box
    return path, count


def test_full_vocabulary_dump(native, full_model):
    import torch
    path, count = full_model
    reader = native.Reader(path)
    checker = oracle.OracleReader(path)
    assert len(reader) == count

    rows = torch.arange(count, dtype=torch.int32, device='cuda')
    out = reader.rows_embedding_device(rows)
    torch.cuda.synchronize()

    # (1) slice-by-slice bit comparison with the checker over the whole vocabulary
    step = 200000
    for start in range(0, count, step):
        stop = min(count, start + step)
        expected = checker.rows_embedding(np.arange(start, stop, dtype=np.uint32))
        assert bits_equal(out[start:stop].cpu().numpy(), expected), (start, stop)

    # (2) every decoded value is one of the <= 16 centroids
    assert torch.unique(out).numel() <= 16

    # (3) idempotence: a second launch into a fresh buffer gives the same bits
    again = reader.rows_embedding_device(rows)
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int32), again.view(torch.int32))
    del again

    # (4) permutation consistency: decode(rows[perm]) == decode(rows)[perm], misses stay zero
    generator = torch.Generator(device='cuda').manual_seed(11)
    perm = torch.randperm(count, device='cuda', generator=generator)[:500000].to(torch.int32)
    perm[::1000] = -1  # 0xFFFFFFFF
    shuffled = reader.rows_embedding_device(perm)
    torch.cuda.synchronize()
    valid = perm >= 0
    assert torch.equal(shuffled[valid].view(torch.int32), out[perm[valid].long()].view(torch.int32))
    assert not bool(shuffled[~valid].any())

    # (5) the whole vocabulary in shuffled order, with and without the caller's hint that it is (order='random': blocks of
    # four wavefronts where a batch this large gets eight by default) -- the same rows either way
    everything = torch.randperm(count, device='cuda', generator=generator).to(torch.int32)
    plain = reader.rows_embedding_device(everything)
    hinted = reader.rows_embedding_device(everything, order='random')
    torch.cuda.synchronize()
    assert torch.equal(plain.view(torch.int32), hinted.view(torch.int32))
    assert torch.equal(plain.view(torch.int32), out[everything.long()].view(torch.int32))


ONE_TILE_WAVES_PER_CU = 28   # memb_hip.hip: what the scalar registers of decode_trained admit (tests/test_isa.py pins it)


def expected_kernel_class(words, words_per_tile, fine_words_per_tile, cus):
    """The kernel a dense batch of `words` words runs with default options (memb_hip.hip planTrained, chooseGeometry;
    DESIGN.md section 5.0 "which kernel owns which configuration") for a row-record model of dim 300 whose tables leave
    every block size the residency the registers allow -- the 2-, 4- and 6-bit models, nibble or byte keys. R = 16
    wavefronts per CU x CUs, one round = 28 x CUs tiles: decode_records_persistent for one round < tiles <= 4 R, else
    decode_trained in blocks of four wavefronts, eight for batches of more than 16 R tiles; with the finer index (more lanes
    per word) while its own tiles fit one round.
    Returns (kernel family, wavefronts per block or None where the rule does not pin them, finer index?)."""
    tiles = (words + words_per_tile - 1) // words_per_tile
    fine_tiles = (words + fine_words_per_tile - 1) // fine_words_per_tile
    resident = 16 * cus
    one_round = ONE_TILE_WAVES_PER_CU * cus
    if fine_tiles <= one_round:
        return 'decode_trained<', 4, True
    if min(2 * resident, one_round) < tiles <= 4 * resident:
        return 'decode_records_persistent<', None, False
    return 'decode_trained<', 8 if tiles > 16 * resident else 4, False


def test_default_path_of_every_batch_size_class(native, full_model):
    """BASELINE.json configs[1] exactly as bench.py and a caller run it -- 100 000 uniformly random rows of the 2.2 M-word
    4-bit model, 1 % misses, seed 11, DEFAULT options -- and batches on both sides of every edge of the kernel-by-batch-size
    rule (tiles = 2R - 1, 2R, 2R + 1, 4R, 4R + 1, 16R, 16R + 1; R = 16 wavefronts x CUs), each bit-compared with the
    checker and with the kernel the context reports for that size. Reference: src/reader.cpp:59-86,
    src/trained_compression.cpp:113-140."""
    import torch
    path, count = full_model
    reader = native.Reader(path)          # default device, default options
    checker = oracle.OracleReader(path, os.cpu_count() or 1)
    assert reader.host_rows_decoded == 0
    words_per_tile = 64 // reader.info()['lanes_per_word']
    fine_words_per_tile = 64 // reader.info(1)['lanes_per_word']
    assert fine_words_per_tile < words_per_tile
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    resident = 16 * cus

    def check(rows, label):
        info = reader.info(len(rows))
        family, waves, fine = expected_kernel_class(len(rows), words_per_tile, fine_words_per_tile, cus)
        assert info['kernel'].startswith(family), (label, len(rows), info['kernel'])
        assert waves is None or info['waves_per_block'] == waves, (label, len(rows), info['waves_per_block'])
        assert (info['lanes_per_word'] > 64 // words_per_tile) == fine, (label, len(rows), info['lanes_per_word'])
        ids = torch.from_numpy(rows.view(np.int32)).cuda()
        out = torch.full((len(rows), 300), 7.0, dtype=torch.float32, device='cuda')
        reader.rows_embedding_device(ids, out=out)
        torch.cuda.synchronize()
        assert bits_equal(out.cpu().numpy(), checker.rows_embedding(rows)), (label, len(rows), info['kernel'])
        return info['kernel']

    # configs[1]: bench.py's batch_rows(count, 100000)
    rng = np.random.default_rng(11)
    rows = rng.integers(0, count, size=100000).astype(np.uint32)
    rows[rng.integers(0, 100000, size=1000)] = 0xFFFFFFFF
    kernel = check(rows, 'configs[1]')
    if count == FULL_VOCAB and resident == 16 * 256:
        assert kernel.startswith('decode_records_persistent<'), kernel   # 12 500 tiles on 256 CUs: between 2R and 4R

    rng = np.random.default_rng(12)
    seen = set()
    one_round = ONE_TILE_WAVES_PER_CU * cus
    # the finer index on one side of its edge (its own tiles fill one round), the usual lanes per word on the other; the
    # one-tile kernel up to one round of the usual tiles, the pipeline from there
    for batch in (one_round * fine_words_per_tile, one_round * fine_words_per_tile + 1,
                  one_round * words_per_tile, one_round * words_per_tile + 1):
        rows = rng.integers(0, count, size=batch).astype(np.uint32)
        rows[rng.integers(0, batch, size=batch // 100)] = 0xFFFFFFFF
        check(rows, 'fine edge {}'.format(batch))
    for multiple in (2, 4, 16):
        for delta in (-1, 0, 1):
            if multiple != 2 and delta == -1:
                continue
            tiles = multiple * resident + delta
            batch = tiles * words_per_tile - (3 if delta == 1 else 0)   # (a ragged last tile on the far side)
            if batch > 4 * count:
                continue
            rows = rng.integers(0, count, size=batch).astype(np.uint32)
            rows[rng.integers(0, batch, size=batch // 100)] = 0xFFFFFFFF
            rows[:batch // 4] = np.arange(batch // 4, dtype=np.uint32) % count   # a key-order run as well
            seen.add(check(rows, '{}R{:+d}'.format(multiple, delta)).split('<')[0])
    assert {'decode_trained', 'decode_records_persistent'} <= seen, seen
    assert reader.host_rows_decoded == 0


def test_block_size_follows_the_order_of_the_batch_before(native, full_model):
    """Very large batches (more than 16 R tiles) run blocks of eight wavefronts for rows in key order and of seven for rows in
    no particular order. The reference's caller cannot say which it brings (src/reader.cpp:49-57 takes words in any order):
    the kernel looks at sixty-four pairs of neighbouring row ids and leaves word for the next launch (noteBatchOrder). The
    rows are the checker's whatever the block size; the caller's hint still overrides."""
    import torch
    path, count = full_model
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    if count <= 16 * 16 * cus * 8:
        pytest.skip('the model is not larger than 16 R tiles')
    reader = native.Reader(path)
    checker = oracle.OracleReader(path, os.cpu_count() or 1)
    rng = np.random.default_rng(41)
    ordered = torch.arange(count, dtype=torch.int32, device='cuda')
    shuffled_host = rng.permutation(count).astype(np.uint32)
    shuffled = torch.from_numpy(shuffled_host.view(np.int32)).cuda()
    picks = np.sort(rng.choice(count, size=20000, replace=False))
    picks_device = torch.from_numpy(picks).cuda()
    out = torch.empty((count, 300), dtype=torch.float32, device='cuda')

    def run(ids, ids_host, **options):
        before = reader.info(count)['waves_per_block']
        out.fill_(7.0)
        reader.rows_embedding_device(ids, out=out, **options)
        torch.cuda.synchronize()
        assert bits_equal(out[picks_device].cpu().numpy(), checker.rows_embedding(np.ascontiguousarray(ids_host[picks])))
        return before, reader.info(count)['waves_per_block']

    in_order = np.arange(count, dtype=np.uint32)
    assert run(ordered, in_order) == (8, 8)            # nothing seen yet: key order is assumed, and confirmed
    assert run(shuffled, shuffled_host) == (8, 7)      # decoded in blocks of eight; the next such batch gets seven
    assert run(shuffled, shuffled_host) == (7, 7)
    assert run(ordered, in_order) == (7, 8)            # and back
    assert run(ordered, in_order, order='random') == (8, 8)   # the caller's hint decides its own launch, not the memory
    # batches at or below 16 R tiles neither look nor are looked at
    small = shuffled[:100000].contiguous()
    reader.rows_embedding_device(small)
    torch.cuda.synchronize()
    assert reader.info(count)['waves_per_block'] == 8 and reader.info(100000)['waves_per_block'] == 4
    # mostly consecutive rows with misses sprinkled in count as key order; runs of eight shuffled among themselves do not
    holes = in_order.copy()
    holes[rng.integers(0, count, size=count // 100)] = 0xFFFFFFFF
    assert run(torch.from_numpy(holes.view(np.int32)).cuda(), holes)[1] == 8
    blocks = (rng.permutation(count // 8)[:, None] * 8 + np.arange(8)[None, :]).reshape(-1).astype(np.uint32)
    blocks = np.concatenate([blocks, np.arange(len(blocks), count, dtype=np.uint32)])
    assert run(torch.from_numpy(blocks.view(np.int32)).cuda(), blocks)[1] == 8   # 7 of 8 pairs consecutive: key order
    assert reader.host_rows_decoded == 0


@pytest.mark.parametrize('bits,words,seed', [(6, 1999995, 1234), (4, FULL_VOCAB, 99)])
def test_block_size_rule_on_the_byte_key_models(native, bits, words, seed):
    """The 6-bit model (BASELINE.json configs[2]) and the byte-key 4-bit one: kernel class, block size and lanes per word
    by batch size as DESIGN.md section 5.0 states them (round 4's text said byte-key models keep blocks of four while the
    6-bit dump ran eight), and rows on both sides of the 16 R edge against the checker (sampled)."""
    import torch
    from memb_amd import synthetic
    os.environ.setdefault('MEMB_SYNTH_DEVICE', '0')
    count = int(os.environ.get('MEMB_TEST_FULL_VOCAB', words))
    path, _ = synthetic.cached_model(count, 300, 'trained', bits, seed=seed)
    reader = native.Reader(path)
    checker = oracle.OracleReader(path, os.cpu_count() or 1)
    assert reader.info()['kernel'].rstrip('>').split(',')[2].strip() == 'false'   # byte keys
    words_per_tile = 64 // reader.info()['lanes_per_word']
    fine_words_per_tile = 64 // reader.info(1)['lanes_per_word']
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    resident = 16 * cus
    one_round = ONE_TILE_WAVES_PER_CU * cus
    rng = np.random.default_rng(bits)
    unordered_seen = False
    for tiles in (1, one_round * fine_words_per_tile // words_per_tile, one_round * fine_words_per_tile // words_per_tile + 1,
                  one_round, one_round + 1, 2 * resident, 2 * resident + 1, 4 * resident, 4 * resident + 1,
                  16 * resident, 16 * resident + 1, (count + words_per_tile - 1) // words_per_tile):
        batch = min(tiles * words_per_tile, 4 * count)
        info = reader.info(batch)
        family, waves, fine = expected_kernel_class(batch, words_per_tile, fine_words_per_tile, cus)
        if waves == 8 and unordered_seen:
            waves = 7   # (the random batch of 16 R + 1 tiles below left word of its order: the next very large batch runs blocks of seven)
        assert info['kernel'].startswith(family), (tiles, info['kernel'])
        assert waves is None or info['waves_per_block'] == waves, (tiles, info['waves_per_block'])
        assert (info['lanes_per_word'] > 64 // words_per_tile) == fine, (tiles, info['lanes_per_word'])
        if tiles in (16 * resident, 16 * resident + 1, 1):
            unordered_seen = unordered_seen or tiles > 16 * resident
            rows = rng.integers(0, count, size=batch).astype(np.uint32)
            rows[rng.integers(0, batch, size=max(1, batch // 100))] = 0xFFFFFFFF
            out = reader.rows_embedding_device(torch.from_numpy(rows.view(np.int32)).cuda())
            picks = np.sort(rng.choice(batch, size=min(batch, 20000), replace=False))
            got = out[torch.from_numpy(picks).cuda()].cpu().numpy()
            assert bits_equal(got, checker.rows_embedding(np.ascontiguousarray(rows[picks]))), tiles


def test_full_dump_of_a_byte_key_4bit_model(native):
    """Off the headline's happy path (bench.py `glove840b-300d-4bit-fullvocab-bytekeys`): the same shape with seed 99, whose
    4-bit code has a 9-bit word -- so no nibble keys: 4-byte table entries, one symbol per byte of the tile. The whole
    dump against the checker in slices, then shuffled rows with misses against the dump."""
    import torch
    from memb_amd import synthetic
    count = int(os.environ.get('MEMB_TEST_FULL_VOCAB', FULL_VOCAB))
    os.environ.setdefault('MEMB_SYNTH_DEVICE', '0')    # (written on the device: the same bytes, tests/test_gpu_writer.py)
    path, _ = synthetic.cached_model(count, 300, 'trained', 4, seed=99)
    reader = native.Reader(path)
    checker = oracle.OracleReader(path, os.cpu_count() or 1)
    info = reader.info(count)
    assert info['max_code_bits'] > 8
    assert info['kernel'].rstrip('>').split(',')[2].strip() == 'false', info['kernel']   # not the nibble-key instantiation
    rows = torch.arange(count, dtype=torch.int32, device='cuda')
    out = reader.rows_embedding_device(rows)
    torch.cuda.synchronize()
    step = 200000
    for start in range(0, count, step):
        stop = min(count, start + step)
        assert bits_equal(out[start:stop].cpu().numpy(), checker.rows_embedding(np.arange(start, stop, dtype=np.uint32))), (start, stop)
    assert torch.unique(out).numel() <= 16
    generator = torch.Generator(device='cuda').manual_seed(12)
    perm = torch.randperm(count, device='cuda', generator=generator)[:600000].to(torch.int32)
    perm[::777] = -1
    shuffled = reader.rows_embedding_device(perm)
    valid = perm >= 0
    assert torch.equal(shuffled[valid].view(torch.int32), out[perm[valid].long()].view(torch.int32))
    assert not bool(shuffled[~valid].any())


def test_full_vocabulary_through_the_word_api(native, full_model):
    # reader[keys()] -- the reference's to_keyed_vectors call (python/memb/reader.py:27-28):
    # word search overlapped with decode and the pinned-ring copy, against the device-resident result
    import torch
    path, count = full_model
    reader = native.Reader(path)
    keys = reader.keys()
    assert len(keys) == count and keys == sorted(keys)

    device_rows = reader.rows_embedding_device(torch.arange(count, dtype=torch.int32, device='cuda'))
    host_rows = reader[keys]
    assert host_rows.shape == (count, 300) and host_rows.dtype == np.float32
    assert bits_equal(host_rows, device_rows.cpu().numpy())
    del host_rows

    # reversed order with unknown words mixed in, into a wider matrix (ReadersUnion's concatenation)
    sample = keys[::-7]
    sample[::100] = ['\x7fnot a word'] * len(sample[::100])
    wide = np.full((len(sample), 310), 3.0, dtype=np.float32)
    reader.batch_embedding_into(sample, wide, 10)
    expected = device_rows[torch.arange(count - 1, -1, -7, device='cuda')].cpu().numpy()
    expected[::100] = 0
    assert bits_equal(wide[:, 10:], expected)
    assert (wide[:, :10] == 3.0).all()


def test_large_batches_from_several_threads(native, full_model):
    # batches above the search/decode overlap threshold from concurrent threads on one Reader: the
    # word-search pool is taken by one caller (the others search on threads of their own), the
    # context serialises the decodes; every caller must get its own rows
    import threading
    import torch
    path, count = full_model
    reader = native.Reader(path)
    keys = reader.keys()
    expected = reader.rows_embedding_device(torch.arange(count, dtype=torch.int32, device='cuda')).cpu().numpy()
    spans = [(0, 300000), (500000, 830000), (count - 280000, count), (1000000, 1001500)]
    results = [None] * len(spans)
    errors = []

    def work(index):
        first, last = spans[index]
        try:
            for _ in range(2):
                results[index] = reader.batch_embedding(keys[first:last])
        except Exception as error:   # pragma: no cover
            errors.append(error)

    threads = [threading.Thread(target=work, args=(index,)) for index in range(len(spans))]
    for thread in threads:
        thread.start()
    for thread in threads:
        thread.join()
    assert not errors, errors
    for (first, last), rows in zip(spans, results):
        assert bits_equal(rows, expected[first:last]), (first, last)


# ---- the other BASELINE.json configurations at their full sizes ----

def _dump_against_checker(native, path, centroid_limit):
    """Full dump in key order: every slice bit-equal to the checker, values are centroids, and a second
    dump of the rows in reverse order gives the same rows (a different tile -> wave assignment)."""
    import torch
    reader = native.Reader(path)
    checker = oracle.OracleReader(path)
    count = len(reader)
    rows = torch.arange(count, dtype=torch.int32, device='cuda')
    out = reader.rows_embedding_device(rows)
    torch.cuda.synchronize()
    step = 200000
    for start in range(0, count, step):
        stop = min(count, start + step)
        expected = checker.rows_embedding(np.arange(start, stop, dtype=np.uint32))
        assert bits_equal(out[start:stop].cpu().numpy(), expected), (start, stop)
    assert torch.unique(out).numel() <= centroid_limit
    # the reverse dump with the OTHER kernel a single model can run (decode_records_persistent where the layout allows)
    reader.set_option('persistent', 2)
    backwards = reader.rows_embedding_device(torch.flip(rows, dims=(0,)).contiguous())
    torch.cuda.synchronize()
    reader.set_option('persistent', 1)
    assert torch.equal(torch.flip(backwards, dims=(0,)).view(torch.int32), out.view(torch.int32))
    return reader, out


def test_fasttext_shaped_6bit_full_dump(native):
    # BASELINE.json configs[2]: 1,999,995 words, trained 6-bit (byte keys, up to 64 centroids)
    from memb_amd import synthetic
    count = int(os.environ.get('MEMB_TEST_FULL_VOCAB', 1999995))
    path, _ = synthetic.cached_model(count, 300, 'trained', 6)
    reader, _ = _dump_against_checker(native, path, 64)
    assert reader.info()['kernel'].startswith('decode_trained<')
    for forced, name in ((2, 'decode_records_persistent<'), (0, 'decode_trained<')):
        reader.set_option('persistent', forced)
        assert reader.info()['kernel'].startswith(name)


def test_glove_shaped_2bit_full_dump_and_its_eight_way_split(native):
    # BASELINE.json configs[3]: the 2,196,017-word 2-bit dump, whole and as the eight slices
    # memb_amd.sharding.shard_range hands to eight GPUs (decoded one after the other on this one)
    import torch
    from memb_amd import synthetic
    from memb_amd.sharding import shard_range
    count = int(os.environ.get('MEMB_TEST_FULL_VOCAB', FULL_VOCAB))
    path, _ = synthetic.cached_model(count, 300, 'trained', 2)
    reader, out = _dump_against_checker(native, path, 4)
    for rank in range(8):
        start, stop = shard_range(count, rank, 8)
        rows = torch.arange(start, stop, dtype=torch.int32, device='cuda')
        piece = reader.rows_embedding_device(rows)
        torch.cuda.synchronize()
        assert torch.equal(piece.view(torch.int32), out[start:stop].view(torch.int32)), rank


def test_union_of_two_full_size_models_500k_words(native):
    # BASELINE.json configs[4]: ReadersUnion concatenate of a GloVe-shaped and a fastText-shaped 4-bit
    # model, 500 000 words of which a quarter is unknown to each model, (n, 600) output merged on the device
    import torch
    from memb_amd import synthetic
    glove = int(os.environ.get('MEMB_TEST_FULL_VOCAB', FULL_VOCAB))
    fasttext = int(os.environ.get('MEMB_TEST_FULL_VOCAB', 1999995))
    path_a, _ = synthetic.cached_model(glove, 300, 'trained', 4)
    path_b, _ = synthetic.cached_model(fasttext, 300, 'trained', 4, seed=4321)
    reader_a, reader_b = native.Reader(path_a), native.Reader(path_b)
    n = min(500000, glove)
    rng = np.random.default_rng(17)
    rows_a = rng.integers(0, len(reader_a), size=n).astype(np.uint32)
    rows_a[rng.random(n) < 0.25] = 0xFFFFFFFF
    rows_b = rng.integers(0, len(reader_b), size=n).astype(np.uint32)
    rows_b[rng.random(n) < 0.25] = 0xFFFFFFFF
    ids = [torch.from_numpy(rows_a.view(np.int32)).cuda(), torch.from_numpy(rows_b.view(np.int32)).cuda()]
    merged = torch.full((n, 600), -1.0, dtype=torch.float32, device='cuda')
    fused = native._memb.union_rows_to_device(
        [reader_a._impl, reader_b._impl], [ids[0].data_ptr(), ids[1].data_ptr()], [0, 300], n,
        merged.data_ptr(), merged.stride(0), torch.cuda.current_stream().cuda_stream, False)
    assert fused, 'two 4-bit models of one geometry share the union kernel'
    torch.cuda.synchronize()
    got = merged.cpu().numpy()
    expected = np.concatenate([oracle.OracleReader(path_a).rows_embedding(rows_a),
                               oracle.OracleReader(path_b).rows_embedding(rows_b)], axis=-1)
    assert bits_equal(got, expected)
    both_missing = (rows_a == 0xFFFFFFFF) & (rows_b == 0xFFFFFFFF)
    assert both_missing.any() and not got[both_missing].any()
    # the same through one launch per reader (column blocks of the one matrix)
    again = torch.full((n, 600), -1.0, dtype=torch.float32, device='cuda')
    reader_a.rows_embedding_device(ids[0], out=again, col_off=0)
    reader_b.rows_embedding_device(ids[1], out=again, col_off=300)
    torch.cuda.synchronize()
    assert torch.equal(again.view(torch.int32), merged.view(torch.int32))
