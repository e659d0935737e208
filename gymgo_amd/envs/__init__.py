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
    """Register 'go-v0' with gym / gymnasium when one of them is importable (gym_go/__init__.py:3-6);
    returns False when neither is installed (they are not in the MI355X image)."""
    for mod in ('gym', 'gymnasium'):
        try:
            registration = __import__(mod + '.envs.registration', fromlist=['register'])
        except ImportError:
            continue
        registration.register(id='go-v0', entry_point='gymgo_amd.envs:GoEnv')
        return True
    return False
