"""gymgo_amd - MI355X-native batched Go environment with the gym_go:go-v0 / GoEnv / gogame API.

The step hot path runs as hand-written HIP kernels (gymgo_amd/csrc/gg_kernels.hip, gfx950) behind the
C-ABI of include/gymgo_amd.h; this package is the thin Python host side (ctypes + torch device
tensors).  There is no CPU fallback.
"""
from gymgo_amd import govars  # noqa: F401


def _register_if_gym_present():
    # gym_go/__init__.py:3-10 registers 'go-v0' on import; do the same when gym / gymnasium exists (importing
    # gymgo_amd.envs pulls in torch, so it only happens when there is something to register with)
    import importlib.util
    for name in ('gym', 'gymnasium'):
        try:
            if importlib.util.find_spec(name) is not None:
                import gymgo_amd.envs  # noqa: F401  (registers on import)
                return True
        except (ImportError, ValueError):
            continue
    return False


_register_if_gym_present()

__all__ = ['govars', 'gogame', 'state_utils', 'envs', 'GoEnv', 'GoVecEnv', 'GoVecEnvParts', 'make', 'register_gym']


def __getattr__(name):
    # lazy: importing gymgo_amd must not require torch/ROCm until the API is used
    import importlib
    if name in ('gogame', 'state_utils', 'envs', '_lib'):
        return importlib.import_module('gymgo_amd.' + name)
    if name in ('GoEnv', 'GoVecEnv', 'GoVecEnvParts', 'make', 'register_gym'):
        return getattr(importlib.import_module('gymgo_amd.envs'), name)
    raise AttributeError(name)
