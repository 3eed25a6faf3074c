"""memb_amd: the batch word-vector lookup path of memb on AMD MI355X (gfx950).

Same Python surface as the reference package (python/memb/__init__.py:1-4):
Reader, ReadersUnion, Builder, available_compression_strategies. Lookups run as
HIP kernels; there is no CPU decode path, so the native extension must be
built (`python build_native.py`) and a HIP device must be present.
"""
import os as _os

try:
    from . import _memb
except ImportError as error:  # fail loudly: nothing here works without the native code
    raise ImportError(
        'memb_amd: the native extension (_memb / libmemb_hip.so) is missing or does not load ({}). '
        'Build it with `python build_native.py` (needs hipcc).'.format(error)) from error

from .builder import Builder
from .reader import Reader
from .readers_union import ReadersUnion

available_compression_strategies = _memb.available_compression_strategies
hip_device_count = _memb.hip_device_count

HIP_LIBRARY_PATH = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), 'libmemb_hip.so')

__all__ = ['Builder', 'Reader', 'ReadersUnion', 'available_compression_strategies', 'hip_device_count']
