"""What the compiler made of the kernels: per kernel the loads / stores by cache policy, LDS-DMA loads,
registers, scratch and LDS, read from the device assembly (no GPU needed).

    python tools/perf/isa.py [substring of the demangled kernel name] [-- extra hipcc flags]

tests/test_isa.py uses kernel_table() to pin facts a source-level reading can get wrong: round 2 shipped
`flag ? *p : __builtin_nontemporal_load(p)`, which LLVM folds into ONE plain load, and reported the
`nt` loads as adopted.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SOURCE = os.path.join(REPO, 'memb_amd', 'csrc', 'memb_hip.hip')
# the flags of build_native.build_hip_library
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fhip-fp32-correctly-rounded-divide-sqrt',
         '-Wno-unused-value', '-Wno-align-mismatch', '-Wno-pass-failed', '-Wno-unused-command-line-argument']


def _tool(name):
    for candidate in (shutil.which(name), '/opt/rocm/bin/' + name, '/opt/rocm/lib/llvm/bin/' + name, '/usr/bin/' + name):
        if candidate and os.path.exists(candidate):
            return candidate
    raise RuntimeError(name + ' not found')


def device_assembly(extra_flags=()):
    with tempfile.TemporaryDirectory() as scratch:
        target = os.path.join(scratch, 'memb_hip.s')
        subprocess.run([_tool('hipcc'), *FLAGS, *extra_flags, '--cuda-device-only', '-S', '-o', target, SOURCE],
                       check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        with open(target) as f:
            return f.read()


def kernel_table(extra_flags=()):
    """{demangled kernel name: facts} for every kernel of libmemb_hip.so"""
    text = device_assembly(extra_flags)
    names = re.findall(r'^\s*\.amdhsa_kernel (\S+)$', text, flags=re.M)
    demangled = subprocess.run([_tool('c++filt')], input='\n'.join(names), stdout=subprocess.PIPE, text=True,
                               check=True).stdout.split('\n')
    table = {}
    for name, pretty in zip(names, demangled):
        start = text.index('\n' + name + ':')
        body = text[start:text.index('.amdhsa_kernel ' + name, start)]
        code = body.split('.section')[0]
        descriptor = text[text.index('.amdhsa_kernel ' + name):]
        descriptor = descriptor[:descriptor.index('.end_amdhsa_kernel')]

        def field(key, where=descriptor):
            match = re.search(r'\.' + key + r'\s+(\d+)', where)
            return int(match.group(1)) if match else None

        metadata = re.search(r'\.name:\s+' + re.escape(name) + r'\n(.*?)\n  - ', text + '\n  - ', flags=re.S)
        meta = metadata.group(1) if metadata else ''
        loads = re.findall(r'^\s*global_load_dwordx4\s.*$', code, flags=re.M)
        stores = re.findall(r'^\s*global_store_dwordx4\s.*$', code, flags=re.M)
        any_loads = re.findall(r'^\s*(?:global|flat|buffer)_load_\w+\s.*$', code, flags=re.M)
        any_stores = re.findall(r'^\s*(?:global|flat|buffer)_store_\w+\s.*$', code, flags=re.M)
        table[pretty] = {
            'symbol': name,
            'load_x4': len(loads),
            'load_x4_nt': sum(1 for line in loads if re.search(r'\bnt\b', line)),
            'store_x4': len(stores),
            'store_x4_nt': sum(1 for line in stores if re.search(r'\bnt\b', line)),
            # of EVERY width (round 4's test counted x4 loads only and missed a non-temporal dword load)
            'load_nt': sum(1 for line in any_loads if re.search(r'\bnt\b', line)),
            'store_nt': sum(1 for line in any_stores if re.search(r'\bnt\b', line)),
            'lds_dma': len(re.findall(r'^\s*(global|buffer)_load_lds_\w+', code, flags=re.M)) +
                       len(re.findall(r'^\s*buffer_load_\w+ .*\blds\b', code, flags=re.M)),
            'scratch_ops': len(re.findall(r'^\s*scratch_(load|store)_', code, flags=re.M)),
            'vgpr': field('amdhsa_next_free_vgpr'),
            'sgpr': field('amdhsa_next_free_sgpr'),
            'accum_offset': field('amdhsa_accum_offset'),
            'private_segment': field('amdhsa_private_segment_fixed_size'),
            'vgpr_count': field('vgpr_count:', meta) if meta else None,
            # (.sgpr_count of the metadata = next_free_sgpr + VCC / flat scratch / XNACK: what the hardware allocates by)
            'sgpr_count': field('sgpr_count:', meta) if meta else None,
        }
    return table


def waves_per_simd(vgpr, sgpr_count=None):
    """MI355X_MICROARCH.md: vector registers -- allocation granule 8, 512 per lane per SIMD; scalar registers
    ('Residency and cooperative launch') -- 800 per SIMD, granule 16 plus 16: .sgpr_count <= 80 -> 8 wavefronts, 81-96 -> 7,
    97-112 -> 6 (the compiler's own `; Occupancy:` line does not know the second rule)."""
    allocated = (vgpr + 7) // 8 * 8
    waves = min(8, 512 // max(allocated, 8))
    if sgpr_count:
        waves = min(waves, 800 // ((sgpr_count + 15) // 16 * 16 + 16))
    return waves


if __name__ == '__main__':
    arguments = sys.argv[1:]
    extra = []
    if '--' in arguments:
        extra = arguments[arguments.index('--') + 1:]
        arguments = arguments[:arguments.index('--')]
    needle = arguments[0] if arguments else ''
    print('%-92s %5s %5s %4s %6s %6s %4s %4s %7s %8s' % (
        'kernel', 'ld.x4', 'nt', 'dma', 'st.x4', 'st.nt', 'vgpr', 'sgpr', 'scratch', 'waves/EU'))   # sgpr = .sgpr_count
    for pretty, facts in sorted(kernel_table(extra).items()):
        if needle in pretty:
            short = pretty.replace('(anonymous namespace)::', '').split('(')[0]
            print('%-92s %5d %5d %4d %6d %6d %4d %4d %7d %8d' % (
                short[:92], facts['load_x4'], facts['load_x4_nt'], facts['lds_dma'], facts['store_x4'], facts['store_x4_nt'],
                facts['vgpr'], facts['sgpr_count'] or facts['sgpr'], facts['private_segment'], waves_per_simd(facts['vgpr'], facts['sgpr_count'])))
