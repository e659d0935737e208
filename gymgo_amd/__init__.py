"""gymgo_amd - MI355X-native batched Go environment with the gym_go:go-v0 / GoEnv / gogame API.

The step hot path runs as hand-written HIP kernels (gymgo_amd/csrc/gg_kernels.hip, gfx950) behind the
C-ABI of include/gymgo_amd.h; this package is the thin Python host side (ctypes + torch device
tensors).  There is no CPU fallback.
"""
from gymgo_amd import govars  # noqa: F401

__all__ = ['govars', 'gogame', 'state_utils', 'envs', 'GoEnv', 'GoVecEnv', 'make', 'register_gym']


def __getattr__(name):
    # lazy: importing gymgo_amd must not require torch/ROCm until the API is used
    import importlib
    if name in ('gogame', 'state_utils', 'envs', '_lib'):
        return importlib.import_module('gymgo_amd.' + name)
    if name in ('GoEnv', 'GoVecEnv', 'make', 'register_gym'):
        return getattr(importlib.import_module('gymgo_amd.envs'), name)
    raise AttributeError(name)
