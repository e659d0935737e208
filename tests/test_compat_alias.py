"""The opt-in `gym_go` alias (compat/gym_go): the reference's own spelling - gym.make('gym_go:go-v0', ...) and
`from gym_go import gogame, govars` (gym_go/__init__.py:3-6, gym_go/tests/test_basics.py:13) - resolves to this package's
classes.  Runs in a subprocess with the stub `gym` of the oracle harness (the image has no gym) and compat/ on the path."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SNIPPET = r'''
import gym, gym_go
from gym_go import gogame, govars
import gymgo_amd, gymgo_amd.gogame, gymgo_amd.envs
env = gym.make('gym_go:go-v0', size=7, komi=2.5, reward_method='heuristic')
assert type(env) is gymgo_amd.envs.GoEnv, type(env)
assert env.size == 7 and env.komi == 2.5 and env.reward_method.value == 'heuristic'
assert gogame is gymgo_amd.gogame and govars is gymgo_amd.govars
import gym_go.gogame, gym_go.state_utils, gym_go.envs.go_env
assert gym_go.gogame is gymgo_amd.gogame
from gym_go.envs.go_env import GoEnv, RewardMethod
assert GoEnv is gymgo_amd.envs.GoEnv
from gym.envs import registration
assert registration.registry['go-v0'] in ('gym_go.envs:GoEnv', 'gymgo_amd.envs:GoEnv'), registration.registry
s = env.reset()
assert s.shape == (govars.NUM_CHNLS, 7, 7) and not s.any()
assert gogame.init_state(5).shape == (6, 5, 5)
print('alias ok', gym_go.BACKEND)
'''


def test_gym_go_alias_resolves_to_this_package():
    env = dict(os.environ)
    env['PYTHONPATH'] = os.pathsep.join([os.path.join(ROOT, 'compat'), os.path.join(ROOT, 'oracle', 'ref_harness', 'stubs'), ROOT])
    out = subprocess.run([sys.executable, '-c', SNIPPET], env=env, cwd='/tmp', capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert 'alias ok gymgo_amd' in out.stdout


def test_alias_is_opt_in():
    """Without compat/ on the path there is no `gym_go` module from this repo (it never shadows a real gym_go)."""
    env = dict(os.environ)
    env['PYTHONPATH'] = ROOT
    out = subprocess.run([sys.executable, '-c', 'import importlib.util as u; print(u.find_spec("gym_go"))'], env=env, cwd='/tmp',
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == 'None', out.stdout + out.stderr
