import importlib
from dataclasses import dataclass, field
from typing import Any, Dict, Optional


@dataclass
class EnvSpec:
    id: str
    entry_point: Any = None
    reward_threshold: Optional[float] = None
    kwargs: Dict[str, Any] = field(default_factory=dict)
    max_episode_steps: Optional[int] = None


registry: Dict[str, EnvSpec] = {}


def register(id, entry_point=None, kwargs=None, reward_threshold=None, max_episode_steps=None, **_):
    registry[id] = EnvSpec(id=id, entry_point=entry_point, kwargs=dict(kwargs or {}),
                           reward_threshold=reward_threshold, max_episode_steps=max_episode_steps)


def make(id, **kwargs):
    spec = registry[id]
    ep = spec.entry_point
    if isinstance(ep, str):
        mod, _, attr = ep.partition(":")
        ep = getattr(importlib.import_module(mod), attr)
    kw = dict(spec.kwargs)
    kw.update(kwargs)
    env = ep(**kw)
    env.spec = spec
    return env
