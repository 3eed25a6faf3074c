"""In-tree build of the native parts.

    libmemb_hip.so   HIP kernels + C ABI (include/memb_hip.h), hipcc, gfx950 only
    _memb.*.so       pybind11 module: C++ Reader / Builder on top of the C ABI
    oracle/...       the CPU checker used by tests (see oracle/README.md)

Everything is built next to its sources so the binaries travel with the tree.
`python build_native.py` builds whatever is out of date (`--force`: everything).
"""
import os
import shutil
import subprocess
import sys
import sysconfig

REPO_DIR = os.path.dirname(os.path.abspath(__file__))
PACKAGE_DIR = os.path.join(REPO_DIR, 'memb_amd')
CSRC = os.path.join(PACKAGE_DIR, 'csrc')
INCLUDE = os.path.join(REPO_DIR, 'include')
ORACLE_DIR = os.path.join(REPO_DIR, 'oracle')
REFERENCE_SRC = '/root/reference/src'

HIP_LIBRARY = os.path.join(PACKAGE_DIR, 'libmemb_hip.so')
EXTENSION = os.path.join(PACKAGE_DIR, '_memb' + sysconfig.get_config_var('EXT_SUFFIX'))
ORACLE_LIBRARY = os.path.join(ORACLE_DIR, 'libmemb_oracle.so')
UNIFORM_EXPR_LIBRARY = os.path.join(ORACLE_DIR, 'libmemb_uniform_expr.so')
REFERENCE_LIBRARY = os.path.join(ORACLE_DIR, '_ref', 'libmemb_ref.so')
CEILINGS_SOURCE = os.path.join(REPO_DIR, 'tools', 'perf', 'ceilings.hip')
CEILINGS_LIBRARY = os.path.join(REPO_DIR, 'tools', 'perf', 'libmemb_ceilings.so')

GPU_ARCH = 'gfx950'


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    stamp = os.path.getmtime(target)
    return any(os.path.getmtime(source) > stamp for source in sources)


def _run(command):
    result = subprocess.run(command, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if result.returncode != 0:
        raise RuntimeError('build step failed: {}\n{}'.format(' '.join(command), result.stdout))
    return result.stdout


def _hipcc():
    for candidate in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if candidate and os.path.exists(candidate):
            return candidate
    raise RuntimeError('hipcc not found: the HIP library cannot be built')


def build_hip_library(force=False, extra_flags=()):
    sources = [os.path.join(CSRC, name) for name in
               ('memb_hip.hip', 'hip_device_common.h', 'hip_trained_kernels.h', 'hip_rowwise_kernels.h',
                'hip_host_path.h', 'hip_encoder.h', 'hip_encoder_kernels.h', 'hip_words.h', 'hip_words_kernels.h', 'worker_pool.h', 'codec.h', 'wire.h')]
    sources.append(os.path.join(INCLUDE, 'memb_hip.h'))
    if force or _newer(HIP_LIBRARY, sources):
        _run([
            _hipcc(), '--offload-arch=' + GPU_ARCH, '-O3', '-std=c++17', '-fPIC', '-shared',
            # uniform dequantisation must stay four separately rounded IEEE operations
            '-ffp-contract=off', '-fhip-fp32-correctly-rounded-divide-sqrt',
            '-Wno-unused-value', '-Wno-align-mismatch', '-Wno-pass-failed', *extra_flags,
            '-o', HIP_LIBRARY, os.path.join(CSRC, 'memb_hip.hip'),
        ])
    return HIP_LIBRARY


def build_extension(force=False):
    import pybind11
    names = ('bindings.cpp', 'reader.cpp', 'builder.cpp', 'compression_strategy.cpp')
    sources = [os.path.join(CSRC, name) for name in names]
    headers = [os.path.join(CSRC, name) for name in
               ('reader.h', 'builder.h', 'compression_strategy.h', 'worker_pool.h', 'codec.h', 'wire.h')]
    headers.append(os.path.join(INCLUDE, 'memb_hip.h'))
    build_hip_library(force)
    if force or _newer(EXTENSION, sources + headers + [HIP_LIBRARY]):
        _run([
            'g++', '-O3', '-std=c++17', '-fPIC', '-shared', '-fvisibility=hidden', '-pthread',
            '-Wall', '-ffp-contract=off',
            '-I' + pybind11.get_include(), '-I' + sysconfig.get_paths()['include'],
            *sources,
            '-L' + PACKAGE_DIR, '-lmemb_hip', '-Wl,-rpath,$ORIGIN',
            '-o', EXTENSION,
        ])
    return EXTENSION


def build_oracle(force=False):
    """The CPU checker (test infrastructure, never imported by the package)."""
    source = os.path.join(ORACLE_DIR, 'memb_oracle.c')
    if not os.path.exists(source):
        return None
    if force or _newer(ORACLE_LIBRARY, [source]):
        _run([
            'gcc', '-O2', '-std=c11', '-fPIC', '-shared', '-pthread', '-Wall', '-ffp-contract=off',
            '-o', ORACLE_LIBRARY, source,
        ])
    # the uniform expression as its own C++ TU, with the reference's flags (CMakeLists.txt:15)
    expression = os.path.join(ORACLE_DIR, 'uniform_expr.cpp')
    if os.path.exists(expression) and (force or _newer(UNIFORM_EXPR_LIBRARY, [expression])):
        _run(['g++', '-std=c++14', '-O3', '-Wall', '-Werror', '-fPIC', '-shared', '-o', UNIFORM_EXPR_LIBRARY, expression])
    return ORACLE_LIBRARY


def build_reference_oracle(force=False):
    """Compile the reference's own std-only decode headers where they lie.

    Only possible where /root/reference exists (not on the GPU box, which uses
    the prebuilt file).
    """
    driver = os.path.join(ORACLE_DIR, 'ref_driver.cpp')
    if not os.path.isdir(REFERENCE_SRC) or not os.path.exists(driver):
        return REFERENCE_LIBRARY if os.path.exists(REFERENCE_LIBRARY) else None
    sources = [driver] + [os.path.join(REFERENCE_SRC, name) for name in
                          ('prefix_code.cpp', 'prefix_code.h', 'huffman_table_decoder.h',
                           'bit_stream_reader.h', 'bit_stream.h')]
    if force or _newer(REFERENCE_LIBRARY, sources):
        os.makedirs(os.path.dirname(REFERENCE_LIBRARY), exist_ok=True)
        _run([
            'g++', '-O3', '-std=c++14', '-fPIC', '-shared', '-pthread', '-I' + REFERENCE_SRC,   # -O3: reference CMakeLists.txt:15
            driver, os.path.join(REFERENCE_SRC, 'prefix_code.cpp'),
            '-o', REFERENCE_LIBRARY,
        ])
    return REFERENCE_LIBRARY


def build_ceilings(force=False):
    """tools/perf/ceilings.hip: the decoder's memory patterns without a decoder (bench.py's
    roofline.box_ceilings). Measurement code: nothing in memb_amd/ loads it."""
    if not os.path.exists(CEILINGS_SOURCE):
        return None
    if force or _newer(CEILINGS_LIBRARY, [CEILINGS_SOURCE]):
        _run([_hipcc(), '--offload-arch=' + GPU_ARCH, '-O3', '-std=c++17', '-fPIC', '-shared',
              '-o', CEILINGS_LIBRARY, CEILINGS_SOURCE])
    return CEILINGS_LIBRARY


def build_all(force=False):
    build_hip_library(force)
    build_extension(force)
    build_oracle(force)
    build_reference_oracle(force)
    build_ceilings(force)


if __name__ == '__main__':
    build_all(force='--force' in sys.argv)
    print('built:', HIP_LIBRARY, EXTENSION)
