import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, 'tests', 'golden')

SIX_WORDS = {  # reference src/tests.cpp:20-27
    'the': [0.0, 1.0, 2.0], 'of': [0.0, -1.0, 2.0], 'th': [2.0, 0.0, 1.0],
    'a': [1.0, 0.0, -2.0], 'tho': [2.0, 0.0, -1.0], 'abc': [-2.0, 0.0, 1.0],
}


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a HIP device (run on the MI355X box)')


def golden_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def bits_equal(a, b):
    """bit-for-bit equality of float32 arrays (NaNs compare by payload)"""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.fixture(scope='session')
def native():
    """Build (if needed) and import the package; the checker is built too."""
    import build_native
    build_native.build_all()
    import memb_amd
    return memb_amd


@pytest.fixture(scope='session')
def model_dir(tmp_path_factory):
    return tmp_path_factory.mktemp('models')


@pytest.fixture(scope='session')
def make_model(native, model_dir):
    """Synthetic model files, cached per parameter set for the whole session."""
    from memb_amd import synthetic
    cache = {}

    def build(count, dim=300, storage='trained', bits=4, seed=1234, distribution='normal'):
        key = (count, dim, storage, bits, seed, distribution)
        if key not in cache:
            path = os.path.join(str(model_dir), 'm_{}_{}_{}_{}_{}_{}.bin'.format(*key))
            words = synthetic.build_file(path, count, dim, storage, bits, seed=seed, distribution=distribution)
            cache[key] = (path, words)
        return cache[key]

    return build


def has_gpu():
    try:
        import memb_amd
        return memb_amd.hip_device_count() > 0
    except Exception:
        return False
