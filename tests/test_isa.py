"""Facts about the compiled kernels that a reading of the source can get wrong (no GPU needed:
hipcc cross-compiles gfx950 to assembly). Round 2 reported non-temporal stream loads as adopted while
the binary held none -- `flag ? *p : __builtin_nontemporal_load(p)` is folded into one plain load."""
import os
import shutil
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'tools', 'perf'))

pytestmark = pytest.mark.skipif(
    not (shutil.which('hipcc') or os.path.exists('/opt/rocm/bin/hipcc')), reason='hipcc not available')


@pytest.fixture(scope='module')
def kernels():
    import isa
    return isa.kernel_table()


def test_no_kernel_holds_a_non_temporal_load(kernels):
    # round 3 made non-temporal stream loads a template argument and measured them for real (+28..48 % on 100 000
    # rows); round 4 removed the variant -- no lookup kernel loads or stores non-temporally, at ANY width (round 4's
    # version of this test counted 16-byte loads only). One named exception: the block kernel of the uniform storage,
    # dequant_uniform<true> (unaligned / very wide outputs; hip_rowwise_kernels.h), reads a row's bytes -- used once per
    # lookup -- with non-temporal dword loads: 500 k random rows 0.167 against 0.178 ms, one per piece in flight.
    # (The writer's quantise_rows does too: it reads every staged vector exactly once.)
    for name, facts in kernels.items():
        if 'decode_' in name or 'dequant_' in name or 'gather_' in name or 'resolve_' in name:
            if 'dequant_uniform<true>' in name:
                assert facts['load_nt'] == 4 and facts['store_nt'] == 0, (name, facts)   # ROWWISE_BATCH loads in flight
            else:
                assert facts['load_nt'] == 0 and facts['store_nt'] == 0, (name, facts)


def test_the_kernel_zoo_is_what_design_md_says(kernels):
    # which kernels exist at all. A single trained model runs decode_trained or, for two to
    # four tiles per 16 wavefronts per CU, decode_records_persistent; unions decode_union_split or decode_trained_union.
    families = {name.split('(anonymous namespace)::')[1].split('<')[0].split('(')[0] for name in kernels}
    # round 5: decode_trained_batches (decode_trained's body over the tiles of several batches), and word -> row on the
    # device: build_word_table at staging, resolve_words per batch
    assert families == {
        'decode_trained', 'decode_trained_batches', 'decode_records_persistent', 'decode_union_split', 'decode_trained_union',
        'dequant_uniform', 'dequant_uniform_tile', 'gather_full',
        'build_word_table', 'resolve_words',
        'repack_streams', 'pack_row_meta',
        'quantise_rows', 'stream_lengths', 'pack_streams'}, families


def test_headline_kernel_resources(kernels):
    # dense fp32 rows (mode 2), nibble keys: the kernel bench.py's headline runs, and the one BASELINE configs[1] runs
    facts = kernels['void (anonymous namespace)::decode_trained<false, 2, true>((anonymous namespace)::TrainedParams)']
    assert facts['private_segment'] == 0 and facts['scratch_ops'] == 0, facts   # no spills
    assert facts['store_x4'] >= 1 and facts['store_x4_nt'] == 0, facts           # plain 16-byte output stores
    facts = kernels['void (anonymous namespace)::decode_records_persistent<false, 2, true>((anonymous namespace)::TrainedParams)']
    assert facts['private_segment'] == 0 and facts['scratch_ops'] == 0, facts
    # six wavefronts per SIMD, 24 per CU (an output burst of 4 pieces for nibble keys: with 5 it took 82 registers and ran five)
    import isa
    assert isa.waves_per_simd(facts['vgpr'], facts['sgpr_count']) == 6, facts


def test_shipped_library_has_no_measurement_switches(kernels):
    # no cache-policy bits on any store (the store-policy experiments of round 2 used inline assembly), and the
    # `debug` option does not exist in a build without -DMEMB_HIP_MEASURE (tests/test_gpu_parity.py::test_boundary_argument_errors asks the library)
    import isa
    text = isa.device_assembly()
    assert ' sc1 nt' not in text and 'off sc0 sc1' not in text


def test_no_kernel_spills(kernels):
    spilled = {name: facts['private_segment'] for name, facts in kernels.items() if facts['private_segment']}
    assert not spilled, spilled


def test_kernels_that_live_on_occupancy_keep_their_registers(kernels):
    # one tile per wavefront: these kernels hide their chain of loads behind other wavefronts. Vector registers would allow
    # eight per SIMD (<= 64; MI355X_MICROARCH.md, register files), but the hardware hands out scalar registers too (800 per
    # SIMD, .sgpr_count rounded up to 16, plus 16: "Residency and cooperative launch"): rounds 1-4 shipped these kernels
    # with 106 of them -- SIX wavefronts per SIMD where the compiler's own occupancy line said eight. hip_trained_kernels.h
    # holds them to a budget now (MEMB_HIP_SGPRS): SEVEN per SIMD, 28 per CU -- memb_hip.hip ONE_TILE_WAVES_PER_CU, which
    # the kernel-by-batch-size rule counts rounds with.
    import isa
    for name, facts in kernels.items():
        if 'decode_union_split<' in name or 'decode_trained<' in name or 'decode_trained_batches<' in name:
            assert isa.waves_per_simd(facts['vgpr']) == 8, (name, facts['vgpr'])
            assert facts['sgpr_count'] is not None and 80 < facts['sgpr_count'] <= 96, (name, facts['sgpr_count'])
            assert isa.waves_per_simd(facts['vgpr'], facts['sgpr_count']) == 7, (name, facts)
