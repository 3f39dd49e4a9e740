"""Minimal stand-in for the `gymnasium` package (pinned 0.29.1 by the reference,
absent from this image and not installable offline).

TEST TOOLING ONLY: lets oracle/refharness import the unmodified reference from
/root/reference to generate golden vectors.  gymnasium contributes no step arithmetic
to the hot path (only space containers and PCG64 seeding used by the samplers), see
SURVEY.md section 8(c).  Never imported by the product (marlon_amd/) or on the GPU box.
"""
from typing import Generic, TypeVar

from . import spaces, error          # noqa: F401
from .spaces import Space            # noqa: F401
from .utils import seeding           # noqa: F401
from .envs import registration       # noqa: F401
from .envs.registration import register, make  # noqa: F401

ObsType = TypeVar("ObsType")
ActType = TypeVar("ActType")


class Env(Generic[ObsType, ActType]):
    metadata: dict = {}
    np_random = None
    action_space = None
    observation_space = None
    spec = None

    @property
    def unwrapped(self):
        return self

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            self.np_random, _ = seeding.np_random(seed)

    def step(self, action):
        raise NotImplementedError

    def close(self):
        return None


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self.action_space = getattr(env, "action_space", None)
        self.observation_space = getattr(env, "observation_space", None)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def step(self, action):
        return self.env.step(action)
