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
GPU_ORDER = ('test_gpu_configs.py', 'test_gpu_deep.py', 'test_gpu_parity.py', 'test_gpu_adversarial.py', 'test_gpu_packed.py',
             'test_gpu_extras.py', 'test_gpu_env.py', 'test_gpu_split.py')


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


def _build_oracle():
    """The C restatement (gcc only): what every CPU-side test needs."""
    import subprocess
    ora_so = os.path.join(ROOT, 'oracle', 'libgg_oracle.so')
    if _stale(ora_so, [os.path.join(ROOT, 'oracle', 'gg_oracle.c')]):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), '-s'])
    return ora_so


@pytest.fixture(scope='session', autouse=True)
def oracle_built():
    """CPU-only sessions (-m "not gpu") need the oracle library and nothing else: no torch import, no hipcc."""
    return _build_oracle()


@pytest.fixture(scope='session')
def native_built(oracle_built):
    """The HIP library is a build product (git-ignored): build it whenever it is missing or older than its sources, so
    that a checkout that carries only tracked files still tests the HIP path - and say which library the session
    loaded.  Requested by every -m gpu test (see _native_for_gpu_tests) and by the C-ABI surface tests; a box without
    hipcc / torch SKIPS those tests instead of failing the oracle / golden ones."""
    import shutil
    csrc = os.path.join(ROOT, 'gymgo_amd', 'csrc')
    lib_so = os.path.join(ROOT, 'gymgo_amd', 'libgymgo_amd.so')
    lib_src = [os.path.join(csrc, f) for f in os.listdir(csrc)] + [os.path.join(ROOT, 'include', 'gymgo_amd.h')]
    if _stale(lib_so, lib_src):
        if shutil.which('hipcc') is None and not os.path.exists('/opt/rocm/bin/hipcc'):
            pytest.skip('libgymgo_amd.so is not built and there is no hipcc on this box')
        import __graft_entry__
        __graft_entry__.build()
    try:
        from gymgo_amd import _lib
    except ImportError as e:      # no torch: the package cannot load its library
        pytest.skip('gymgo_amd._lib needs torch (%s)' % e)
    L = _lib.lib()
    sys.stderr.write('\n[gymgo_amd] native library %s (ABI %d), oracle %s\n' % (_lib.LIB_PATH, L.gg_version(), oracle_built))
    return L


@pytest.fixture(autouse=True)
def _native_for_gpu_tests(request):
    if request.node.get_closest_marker('gpu') is not None:
        request.getfixturevalue('native_built')


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
