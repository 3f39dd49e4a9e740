"""Container-only stand-ins for gymnasium.spaces (no step arithmetic)."""
from collections import OrderedDict

import numpy as np

from . import space                     # noqa: F401
from .space import Space


class Discrete(Space):
    def __init__(self, n, seed=None, start=0):
        super().__init__((), np.int64, seed)
        self.n = int(n)
        self.start = int(start)

    def sample(self, mask=None):
        return int(self.start + self.np_random.integers(self.n))

    def contains(self, x):
        return self.start <= int(x) < self.start + self.n

    def __eq__(self, other):
        return isinstance(other, Discrete) and self.n == other.n and self.start == other.start


class MultiDiscrete(Space):
    def __init__(self, nvec, dtype=np.int64, seed=None):
        self.nvec = np.array(nvec, dtype=dtype, copy=True)
        super().__init__(self.nvec.shape, dtype, seed)

    def sample(self, mask=None):
        return (self.np_random.random(self.nvec.shape) * self.nvec).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.nvec.shape and bool(np.all(x >= 0) and np.all(x < self.nvec))

    def __eq__(self, other):
        return isinstance(other, MultiDiscrete) and np.array_equal(self.nvec, other.nvec)


class MultiBinary(Space):
    def __init__(self, n, seed=None):
        if isinstance(n, (list, tuple, np.ndarray)):
            self.n = tuple(int(i) for i in n)
            shape = self.n
        else:
            self.n = int(n)
            shape = (self.n,)
        super().__init__(shape, np.int8, seed)

    def sample(self, mask=None):
        return self.np_random.integers(0, 2, size=self.shape, dtype=np.int8)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all((x == 0) | (x == 1)))

    def __eq__(self, other):
        return isinstance(other, MultiBinary) and self.n == other.n


class Tuple(Space):
    def __init__(self, spaces, seed=None):
        self.spaces = tuple(spaces)
        super().__init__(None, None, seed)

    def sample(self, mask=None):
        return tuple(s.sample() for s in self.spaces)

    def contains(self, x):
        return len(x) == len(self.spaces)

    def __len__(self):
        return len(self.spaces)

    def __getitem__(self, i):
        return self.spaces[i]

    def __eq__(self, other):
        return isinstance(other, Tuple) and self.spaces == other.spaces


class Dict(Space):
    def __init__(self, spaces=None, seed=None, **spaces_kwargs):
        if spaces is None:
            spaces = spaces_kwargs
        if isinstance(spaces, dict) and not isinstance(spaces, OrderedDict):
            spaces = OrderedDict(spaces.items())
        self.spaces = OrderedDict(spaces)
        super().__init__(None, None, seed if not isinstance(seed, dict) else None)

    def sample(self, mask=None):
        return OrderedDict((k, s.sample()) for k, s in self.spaces.items())

    def contains(self, x):
        return isinstance(x, dict) and set(x.keys()) == set(self.spaces.keys())

    def __getitem__(self, key):
        return self.spaces[key]

    def __iter__(self):
        return iter(self.spaces)

    def __len__(self):
        return len(self.spaces)

    def keys(self):
        return self.spaces.keys()

    def to_jsonable(self, sample_n):
        return {k: [s[k] for s in sample_n] for k in self.spaces}

    def from_jsonable(self, sample_n):
        n = len(next(iter(sample_n.values())))
        return [OrderedDict((k, sample_n[k][i]) for k in self.spaces) for i in range(n)]

    def __eq__(self, other):
        return isinstance(other, Dict) and self.spaces == other.spaces
