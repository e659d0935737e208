"""Environments: GoEnv (the reference's single-game env, gym_go/envs/go_env.py) and GoVecEnv (the batched form).

Like `import gym_go` (gym_go/__init__.py:3-10), importing this package registers 'go-v0' with gym / gymnasium when
one of them is importable, so `gym.make('gymgo_amd:go-v0', size=19)` works; `make()` below does the same without gym.
'go-extrahard-v0' is an empty class in the reference (gym_go/envs/go_extrahard_env.py) and is not provided.
"""
from gymgo_amd.envs import spaces  # noqa: F401
from gymgo_amd.envs.go_env import GoEnv, RewardMethod  # noqa: F401
from gymgo_amd.envs.vec_env import GoVecEnv, GoVecEnvParts  # noqa: F401

ENV_IDS = {'go-v0': GoEnv}


def make(spec, **kwargs):
    """gym.make-style constructor that works without `gym` installed:
    make('gym_go:go-v0', size=19) / make('gymgo_amd:go-v0', ...) / make('go-v0', ...)."""
    env_id = spec.split(':')[-1]
    if env_id not in ENV_IDS:
        raise KeyError('unknown environment id %r' % spec)
    return ENV_IDS[env_id](**kwargs)


ENTRY_POINT = 'gymgo_amd.envs:GoEnv'
GYM_IDS = ('go-v0', 'gymgo_amd/go-v0')     # the reference's bare id + one that cannot collide with `import gym_go`


def _registered_entry_point(env_id):
    """The entry point gym / gymnasium currently holds for `env_id` (None if unknown)."""
    try:
        registry = __import__(spaces.gym.__name__ + '.envs.registration', fromlist=['registry']).registry
        spec = registry.get(env_id) if hasattr(registry, 'get') else registry.env_specs.get(env_id)   # old gym: EnvRegistry
        return spec if spec is None or isinstance(spec, str) else getattr(spec, 'entry_point', None)
    except Exception:
        return None


def register_gym():
    """Register 'go-v0' (gym_go/__init__.py:3-6) and the collision-free 'gymgo_amd/go-v0' with gym / gymnasium.  Returns
    False when neither is installed (they are not in the MI355X image).  The bare id is shared with the reference
    package: if `gym_go` was imported first (or the registry refuses to re-register) the registry keeps ITS entry
    point, and `gym.make('gymgo_amd:go-v0')` would then build the CPU reference env - that is checked and reported
    (RuntimeWarning + GYM_REGISTRATION['go-v0'] = False) instead of silently swallowed; 'gymgo_amd/go-v0' and
    gymgo_amd.envs.make() always give this package's GoEnv.  Runs once on import."""
    import warnings
    if spaces.gym is None:
        return False
    try:
        registration = __import__(spaces.gym.__name__ + '.envs.registration', fromlist=['register'])
    except Exception:
        return False
    for env_id in GYM_IDS:
        try:
            registration.register(id=env_id, entry_point=ENTRY_POINT)
        except Exception:   # e.g. gym's "Cannot re-register id": decided by the check below
            pass
        held = _registered_entry_point(env_id)
        ok = held == ENTRY_POINT or held is GoEnv
        GYM_REGISTRATION[env_id] = ok
        if not ok and env_id == 'go-v0':
            warnings.warn("gym id 'go-v0' is held by %r, not by gymgo_amd (was gym_go imported first?): "
                          "use gym.make('gymgo_amd/go-v0') or gymgo_amd.envs.make('go-v0')" % (held,), RuntimeWarning)
    return GYM_REGISTRATION.get('go-v0', False) or GYM_REGISTRATION.get('gymgo_amd/go-v0', False)


GYM_REGISTRATION = {}
REGISTERED_WITH_GYM = register_gym()
