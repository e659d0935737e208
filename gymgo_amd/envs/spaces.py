"""observation_space / action_space of GoEnv (gym_go/envs/go_env.py:35-37).

When `gym` or `gymnasium` is importable the real `spaces.Box` / `spaces.Discrete` and `Env` base class are used;
the MI355X image has neither, so minimal stand-ins with the attributes and methods self-play loops touch
(`.n`, `.shape`, `.dtype`, `.low`, `.high`, `.sample()`, `.contains()`, `in`) keep `env.action_space.n` /
`env.action_space.sample()` / `env.observation_space.shape` working either way.
"""
import numpy as np


def _find_gym():
    for name in ('gym', 'gymnasium'):
        try:
            mod = __import__(name)
            __import__(name + '.spaces')
            return mod
        except Exception:   # ImportError, or a gym too old / too broken to import under this NumPy
            continue
    return None


gym = _find_gym()


class _Space:
    def __contains__(self, x):
        return self.contains(x)

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)
        return [seed]

    @property
    def np_random(self):
        if getattr(self, '_rng', None) is None:
            self._rng = np.random.default_rng()
        return self._rng


class _Discrete(_Space):
    """{0, 1, ..., n-1}."""

    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)

    def sample(self):
        return int(self.np_random.integers(self.n))

    def contains(self, x):
        try:
            i = int(x)
        except (TypeError, ValueError):
            return False
        return i == x and 0 <= i < self.n

    def __repr__(self):
        return 'Discrete(%d)' % self.n

    def __eq__(self, other):
        return getattr(other, 'n', None) == self.n and getattr(other, 'shape', None) == ()


class _Box(_Space):
    """Box(low, high, shape): scalar bounds broadcast over `shape`, float32 like gym's default."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        self.shape = tuple(shape) if shape is not None else np.shape(low)
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)

    def sample(self):
        return self.np_random.uniform(self.low, self.high, size=self.shape).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low)) and bool(np.all(x <= self.high))

    def __repr__(self):
        return 'Box(%s, %s, %s, %s)' % (self.low.min(), self.high.max(), self.shape, self.dtype)

    def __eq__(self, other):
        return (getattr(other, 'shape', None) == self.shape and np.array_equal(getattr(other, 'low', None), self.low)
                and np.array_equal(getattr(other, 'high', None), self.high))


if gym is not None:
    Box, Discrete, Env = gym.spaces.Box, gym.spaces.Discrete, gym.Env
else:
    Box, Discrete = _Box, _Discrete

    class Env:   # what gym.Env gives GoEnv when gym is absent: the attribute names, nothing else
        metadata = {}
        reward_range = (-float('inf'), float('inf'))
        spec = None

        @property
        def unwrapped(self):
            return self

        def seed(self, seed=None):
            return [seed]
