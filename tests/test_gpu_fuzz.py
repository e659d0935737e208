"""-m gpu: a few seconds of tests/fuzz_parity.py (differential fuzz of the HIP path against the oracle) in the suite; run the
script itself for longer sessions and other seeds."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize('cus', [None, 16, 4])
def test_a_few_seconds_of_differential_fuzz(cus):
    env = dict(os.environ)
    env.pop('GYMGO_AMD_CUS', None)
    if cus:
        env['GYMGO_AMD_CUS'] = str(cus)     # batches of a few thousand (16) / a thousand (4) games then take the big-batch kernels, k_rollout5 among them
    p = subprocess.run([sys.executable, os.path.join(HERE, 'fuzz_parity.py'), '6', '20260928'], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    assert 'fuzz ok' in p.stdout
