"""Opt-in alias: the reference's own spelling on the MI355X backend.

Every caller of the reference writes `gym.make('gym_go:go-v0', size=...)` and `from gym_go import gogame, govars`
(gym_go/__init__.py:3-6, gym_go/tests/test_basics.py:13, README "Low level API").  With THIS directory's parent
(`<repo>/compat`) put on PYTHONPATH by the user, those lines run unchanged and get gymgo_amd's classes and modules:

    PYTHONPATH=<repo>/compat:<repo> python self_play.py

It is never installed automatically and must not be combined with the real `gym_go` (whichever comes first on the path
wins the name).  Nothing here computes anything: the modules below ARE gymgo_amd's.
"""
import sys

import gymgo_amd
from gymgo_amd import gogame, govars, state_utils  # noqa: F401  (`from gym_go import gogame, govars`)
from gymgo_amd import envs as _amd_envs

from . import envs  # noqa: F401,E402  (gym_go.envs:GoEnv)

# `import gym_go.gogame` / `from gym_go.state_utils import ...` resolve to the same module objects
for _name, _mod in (('gogame', gogame), ('govars', govars), ('state_utils', state_utils)):
    sys.modules.setdefault(__name__ + '.' + _name, _mod)

ENTRY_POINT = 'gym_go.envs:GoEnv'
BACKEND = gymgo_amd.__name__


def _register():
    """gym_go/__init__.py:3-6: id 'go-v0' -> gym_go.envs:GoEnv (this package's GoEnv).  Importing gymgo_amd.envs has
    already registered the bare id with the same class under its own module path; gym versions that refuse to re-register
    keep that entry, which builds the identical class."""
    gym = _amd_envs.spaces.gym
    if gym is None:
        return False
    try:
        registration = __import__(gym.__name__ + '.envs.registration', fromlist=['register'])
    except Exception:
        return False
    try:
        reg = getattr(registration, 'registry', None)
        if isinstance(reg, dict):
            reg.pop('go-v0', None)
        registration.register(id='go-v0', entry_point=ENTRY_POINT)
    except Exception:
        pass
    return True


REGISTERED_WITH_GYM = _register()
