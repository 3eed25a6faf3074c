"""memb_amd: the batch word-vector lookup path of memb on AMD MI355X (gfx950).

Same Python surface as the reference package (python/memb/__init__.py:1-4):
Reader, ReadersUnion, Builder, available_compression_strategies. Lookups run as
HIP kernels: the native extension must be built (`python build_native.py`) and a
HIP device must be present -- a reader never falls back to the CPU on its own.
The reference's host path exists only on request: Reader(path, device='cpu')
for hosts without a GPU, host_below=N to keep tiny host batches off the GPU.
"""
import os as _os


def _preload_hip_runtime():
    """One HIP runtime per process.

    PyTorch-ROCm wheels bundle their own libamdhip64.so (soname libamdhip64.so.7,
    like the system one in /opt/rocm). If libmemb_hip.so pulled in the system
    copy first and torch were imported later, the process would hold two HIP/HSA
    runtimes: torch then finds no GPU, and streams or events made by one runtime
    mean nothing to the other. Loading torch's copy first makes the dynamic
    loader resolve libmemb_hip.so's dependency to it (same soname). Without
    torch the system runtime is used. MEMB_HIP_RUNTIME overrides the choice.
    """
    import ctypes
    import importlib.util
    candidates = []
    if _os.environ.get('MEMB_HIP_RUNTIME'):
        candidates.append(_os.environ['MEMB_HIP_RUNTIME'])
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.submodule_search_locations:
        candidates.append(_os.path.join(list(spec.submodule_search_locations)[0], 'lib', 'libamdhip64.so'))
    for candidate in candidates:
        if _os.path.exists(candidate):
            try:
                ctypes.CDLL(candidate, mode=ctypes.RTLD_GLOBAL)
                return candidate
            except OSError:
                continue
    return None


HIP_RUNTIME_PRELOADED = _preload_hip_runtime()

try:
    from . import _memb
except ImportError as error:  # fail loudly: nothing here works without the native code
    raise ImportError(
        'memb_amd: the native extension (_memb / libmemb_hip.so) is missing or does not load ({}). '
        'Build it with `python build_native.py` (needs hipcc).'.format(error)) from error

from .builder import Builder
from .reader import Reader
from .readers_union import ReadersUnion
from .sharding import ShardedReader

available_compression_strategies = _memb.available_compression_strategies
hip_device_count = _memb.hip_device_count

HIP_LIBRARY_PATH = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), 'libmemb_hip.so')

__all__ = ['Builder', 'Reader', 'ReadersUnion', 'ShardedReader', 'available_compression_strategies', 'hip_device_count']
