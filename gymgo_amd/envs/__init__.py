"""Environments: GoEnv (the reference's single-game env, gym_go/envs/go_env.py) and GoVecEnv (the batched form).

Like `import gym_go` (gym_go/__init__.py:3-10), importing this package registers 'go-v0' with gym / gymnasium when
one of them is importable, so `gym.make('gymgo_amd:go-v0', size=19)` works; `make()` below does the same without gym.
'go-extrahard-v0' is an empty class in the reference (gym_go/envs/go_extrahard_env.py) and is not provided.
"""
from gymgo_amd.envs import spaces  # noqa: F401
from gymgo_amd.envs.go_env import GoEnv, RewardMethod  # noqa: F401
from gymgo_amd.envs.vec_env import GoVecEnv  # noqa: F401

ENV_IDS = {'go-v0': GoEnv}


def make(spec, **kwargs):
    """gym.make-style constructor that works without `gym` installed:
    make('gym_go:go-v0', size=19) / make('gymgo_amd:go-v0', ...) / make('go-v0', ...)."""
    env_id = spec.split(':')[-1]
    if env_id not in ENV_IDS:
        raise KeyError('unknown environment id %r' % spec)
    return ENV_IDS[env_id](**kwargs)


def register_gym():
    """Register 'go-v0' with gym / gymnasium (gym_go/__init__.py:3-6); False when neither is installed (they are
    not in the MI355X image) or the id is registered already.  Runs once on import."""
    if spaces.gym is None:
        return False
    try:
        registration = __import__(spaces.gym.__name__ + '.envs.registration', fromlist=['register'])
        registration.register(id='go-v0', entry_point='gymgo_amd.envs:GoEnv')
        return True
    except Exception:   # e.g. gym's "Cannot re-register id" when imported twice under different names
        return False


REGISTERED_WITH_GYM = register_gym()
