"""Load the UNMODIFIED reference hot path from /root/reference (this container only).

TEST TOOLING: used by gen_golden.py to produce the committed fixtures under tests/golden/.
Nothing here runs on the GPU box (the reference does not travel) and nothing here is
imported by the product package.  Recipe follows SURVEY.md Appendix D:
  1. stand-ins for the three absent third-party packages (gymnasium, boolean.py, IPython)
     are put on sys.path (oracle/refharness/standins);
  2. an empty `cyberbattle` package whose __path__ is the reference directory is
     pre-registered so cyberbattle/__init__.py (which pulls agents -> progressbar) is skipped;
  3. numpy.can_cast is shimmed to accept Python ints (numpy>=2 raises; env.py:206-214).
"""
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("MCBS_REFERENCE_ROOT", "/root/reference")
_CBS = os.path.join(REFERENCE_ROOT, "src", "CyberBattleSim", "cyberbattle")
_STANDINS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "standins")

_loaded = None


def available() -> bool:
    return os.path.isdir(_CBS)


def load():
    """Return a namespace with the reference modules (env, model, actions, defender, chain, toyctf)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError(f"reference not present at {REFERENCE_ROOT}")
    if _STANDINS not in sys.path:
        sys.path.insert(0, _STANDINS)

    _cc = np.can_cast

    def can_cast(x, t, *a, **k):
        if isinstance(x, (int, np.integer)) and not isinstance(x, bool):
            info = np.iinfo(t)
            return info.min <= int(x) <= info.max
        return _cc(x, t, *a, **k)

    np.can_cast = can_cast

    pkg = types.ModuleType("cyberbattle")
    pkg.__path__ = [_CBS]
    sys.modules["cyberbattle"] = pkg

    from cyberbattle._env import cyberbattle_env as env
    from cyberbattle._env import defender
    from cyberbattle._env.cyberbattle_chain import CyberBattleChain
    from cyberbattle._env.cyberbattle_toyctf import CyberBattleToyCtf
    from cyberbattle.simulation import actions, model, commandcontrol
    from cyberbattle.samples.chainpattern import chainpattern
    from cyberbattle.samples.toyctf import toy_ctf

    # marlon's wrappers import a plotly symbol that plotly 6 no longer ships (attack_wrapper.py:6); rendering only
    if "plotly.missing_ipywidgets" not in sys.modules:
        shim = types.ModuleType("plotly.missing_ipywidgets")
        shim.FigureWidget = type("FigureWidget", (), {})
        sys.modules["plotly.missing_ipywidgets"] = shim
    if REFERENCE_ROOT not in sys.path:
        sys.path.append(REFERENCE_ROOT)          # `marlon.baseline_models.env_wrappers.*`

    ns = types.SimpleNamespace(env=env, defender=defender, actions=actions, model=model,
                               commandcontrol=commandcontrol, chainpattern=chainpattern, toy_ctf=toy_ctf,
                               CyberBattleChain=CyberBattleChain, CyberBattleToyCtf=CyberBattleToyCtf)
    _loaded = ns
    return ns
