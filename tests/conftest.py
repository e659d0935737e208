import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')

# -m gpu runs stop at the first failure (-x): the parity tests of the BASELINE configs go first - config 3 (19x19 x
# 65 536 rollouts), config 2 (9x9 x 4 096), config 5 (children), then every other oracle / golden comparison, then the
# env and host-surface tests.  Within a file the source order is kept.
GPU_ORDER = ('test_gpu_configs.py', 'test_gpu_parity.py', 'test_gpu_adversarial.py', 'test_gpu_packed.py',
             'test_gpu_extras.py', 'test_gpu_env.py')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.basename(str(item.fspath))
        return GPU_ORDER.index(name) if name in GPU_ORDER else len(GPU_ORDER)
    items.sort(key=rank)   # stable: source order inside a file


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


@pytest.fixture(scope='session', autouse=True)
def native_built():
    """The .so files are build products (git-ignored): build them whenever they are missing or older than their
    sources, so that a checkout that carries only tracked files still tests the HIP path - and say which library the
    session loaded."""
    csrc = os.path.join(ROOT, 'gymgo_amd', 'csrc')
    lib_so = os.path.join(ROOT, 'gymgo_amd', 'libgymgo_amd.so')
    lib_src = [os.path.join(csrc, f) for f in os.listdir(csrc)] + [os.path.join(ROOT, 'include', 'gymgo_amd.h')]
    ora_so = os.path.join(ROOT, 'oracle', 'libgg_oracle.so')
    ora_src = [os.path.join(ROOT, 'oracle', 'gg_oracle.c')]
    if _stale(lib_so, lib_src) or _stale(ora_so, ora_src):
        import __graft_entry__
        __graft_entry__.build()
    from gymgo_amd import _lib
    L = _lib.lib()
    sys.stderr.write('\n[gymgo_amd] native library %s (ABI %d), oracle %s\n' % (_lib.LIB_PATH, L.gg_version(), ora_so))
    return L


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'))
    return load


@pytest.fixture(scope='session')
def scripted_cases():
    import json
    return json.load(open(os.path.join(GOLDEN, 'scripted_cases.json')))['cases']
