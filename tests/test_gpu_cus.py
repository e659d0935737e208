"""-m gpu: the library sized for another number of compute units (GYMGO_AMD_CUS) computes the same results.

Grids, batch splits by wave age, the take-over sizes of the sixteen-board kernels and the choice between the two-board and
the multi-ply rollout kernel all derive from the CU count; the tuning was done on a 256-CU device.  Each setting runs in
a process of its own (the variable is read once) over the entry points whose launch shape depends on it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import hashlib, json, sys
sys.path.insert(0, %r)
import torch
from gymgo_amd import gogame, _lib
from gymgo_amd.envs import GoVecEnv
out = {'cus': _lib.lib().gg_device_cus()}
def dig(*ts):
    h = hashlib.sha1()
    for t in ts:
        h.update(t.contiguous().cpu().numpy().tobytes())
    return h.hexdigest()[:16]
for N, B in ((19, 20000), (13, 9000), (9, 6000)):
    st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 5)
    ch = B // 8
    for g in range(8):
        gogame.batch_rollout(st[g * ch:(g + 1) * ch], rng[g * ch:(g + 1) * ch], 11 * g + 3, True)
    out['mix%%d' %% N] = dig(st, rng)
    acts = gogame.batch_sample_actions(st, rng.clone())
    nxt, status = gogame.batch_next_states(st, acts, check=False)
    out['next%%d' %% N] = dig(nxt, status)
    out['mask%%d' %% N] = dig(gogame._invalid_mask_dev(st))
    b, w = gogame.batch_areas(st)
    out['areas%%d' %% N] = dig(b, w)
    s2, r2 = st.clone(), rng.clone()
    gogame.batch_rollout(s2, r2, 1, True)
    out['ply%%d' %% N] = dig(s2, r2)
    gogame.batch_rollout(s2, r2, 37, True)
    out['fused%%d' %% N] = dig(s2, r2)
    e = gogame.batch_env_step(st.clone(), None, rng.clone(), 0.5, 'heuristic', True)
    out['env%%d' %% N] = dig(*e)
    kids = gogame.batch_children(st[:96])
    out['kids%%d' %% N] = dig(kids)
    tr = gogame.batch_track(st)
    out['track%%d' %% N] = dig(tr)
    env = GoVecEnv(B, N, seed=9); env.states = st
    for _ in range(3):
        o = env.step()
    out['venv%%d' %% N] = dig(*o, env.tracked, env.rng)
    out['sym%%d' %% N] = dig(gogame.batch_symmetry(st[:512]))
print('DIGESTS ' + json.dumps(out))
''' % ROOT


def _run(cus):
    env = dict(os.environ)
    env.pop('GYMGO_AMD_CUS', None)
    if cus:
        env['GYMGO_AMD_CUS'] = str(cus)
    p = subprocess.run([sys.executable, '-c', SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith('DIGESTS ')][-1]
    return json.loads(line[len('DIGESTS '):])


def test_results_do_not_depend_on_the_cu_count():
    base = _run(None)
    assert base['cus'] == torch.cuda.get_device_properties(0).multi_processor_count
    for cus in (8, 60, 304, 1024):
        got = _run(cus)
        assert got.pop('cus') == cus
        want = dict(base); want.pop('cus')
        assert got == want, {k: (got[k], want[k]) for k in want if got[k] != want[k]}
