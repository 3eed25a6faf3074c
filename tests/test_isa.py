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


def _persistent(kernels, nt):
    wanted = ', true>' if nt else ', false>'
    found = {name: facts for name, facts in kernels.items()
             if 'decode_trained_persistent<' in name and name.split('(anonymous namespace)::TrainedParams')[0].rstrip('(').endswith(wanted)}
    assert found, 'no decode_trained_persistent<..., {}> kernel in the device code'.format('true' if nt else 'false')
    return found


def test_nt_variant_of_the_persistent_kernel_holds_nt_loads(kernels):
    # prologue (3 index records + 2 x 4 stream pieces) and loop (1 + 4): every one of them non-temporal
    for name, facts in _persistent(kernels, nt=True).items():
        assert facts['load_x4_nt'] >= 10, (name, facts)
    for name, facts in _persistent(kernels, nt=False).items():
        assert facts['load_x4_nt'] == 0, (name, facts)


def test_headline_kernel_resources(kernels):
    # dense fp32 rows (mode 2), nibble keys: the kernel bench.py's headline runs
    for nt in (False, True):
        name = 'void (anonymous namespace)::decode_trained_persistent<false, 2, true, {}>((anonymous namespace)::TrainedParams)'.format(
            'true' if nt else 'false')
        facts = kernels[name]
        assert facts['private_segment'] == 0 and facts['scratch_ops'] == 0, facts   # no spills
        assert facts['vgpr'] <= 128, facts                                            # 4 waves per SIMD
        assert facts['store_x4'] >= 1 and facts['store_x4_nt'] == 0, facts           # plain 16-byte output stores


def test_shipped_library_has_no_measurement_switches(kernels):
    # the store-policy experiments use inline `global_store_dwordx4 ... sc1/nt`: none of it may survive in a
    # build without -DMEMB_HIP_MEASURE
    import isa
    text = isa.device_assembly()
    assert ' sc1 nt' not in text and 'off sc0 sc1' not in text
    assert all(facts['store_x4_nt'] == 0 for facts in kernels.values())


def test_no_kernel_spills(kernels):
    spilled = {name: facts['private_segment'] for name, facts in kernels.items() if facts['private_segment']}
    assert not spilled, spilled


def test_kernels_that_live_on_occupancy_keep_their_registers(kernels):
    # one tile per wavefront: these kernels hide their chain of loads behind other wavefronts, eight per SIMD
    # (<= 64 VGPRs; MI355X_MICROARCH.md, register files) -- a rewrite that costs registers costs them that
    import isa
    for name, facts in kernels.items():
        if 'decode_union_split<' in name or ('decode_trained<' in name and ', 2, true>' in name):
            assert isa.waves_per_simd(facts['vgpr']) == 8, (name, facts['vgpr'])
