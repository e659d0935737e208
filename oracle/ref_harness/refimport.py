"""Import the read-only reference (`/root/reference/gym_go`) inside THIS container.

TEST INFRASTRUCTURE ONLY - never imported by the product path, never shipped to the GPU box
(`/root/reference` does not exist there).  Recipe from SURVEY.md section 8(c):
  * stub `gym` / `pyglet` packages on sys.path,
  * `numpy.int = int` (gym_go/gogame.py:250 uses the alias removed in NumPy 1.24).
"""
import os
import sys
import warnings

REFERENCE_ROOT = os.environ.get('GYMGO_REFERENCE', '/root/reference')
_STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'stubs')


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'gym_go'))


def load():
    """Returns (gym, gogame, govars, state_utils) of the real reference."""
    if not available():
        raise RuntimeError('reference not present at %s' % REFERENCE_ROOT)
    import numpy as np
    if not hasattr(np, 'int'):
        np.int = int
    for p in (REFERENCE_ROOT, _STUBS):
        if p not in sys.path:
            sys.path.insert(0, p)
    sys.dont_write_bytecode = True
    warnings.filterwarnings('ignore', category=DeprecationWarning)
    import gym
    import gym_go  # noqa: F401  (registers go-v0)
    from gym_go import gogame, govars, state_utils
    return gym, gogame, govars, state_utils
