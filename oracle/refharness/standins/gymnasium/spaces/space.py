from ..utils import seeding


class Space:
    def __init__(self, shape=None, dtype=None, seed=None):
        self._shape = None if shape is None else tuple(shape)
        self.dtype = dtype
        self._np_random = None
        if seed is not None and not isinstance(seed, dict):
            self.seed(seed)

    @property
    def shape(self):
        return self._shape

    @property
    def np_random(self):
        if self._np_random is None:
            self.seed()
        return self._np_random

    def seed(self, seed=None):
        self._np_random, seed = seeding.np_random(seed)
        return [seed]

    def sample(self, mask=None):
        raise NotImplementedError

    def contains(self, x):
        raise NotImplementedError

    def __contains__(self, x):
        return self.contains(x)
