"""The CPU oracle replays every golden trace captured from the imported reference, bit for bit:
rewards (clamped and raw), termination, step counts, availability (fp64 bit pattern), every numeric
observation field, discovery order and credential-cache order."""
import numpy as np
import pytest

from oracle.oracle import Oracle
from tests import parity


class OracleStepper:
    def __init__(self, topo, spec):
        self.o = Oracle(topo, spec)

    def reset_observation(self, fields):
        return self.o.observe(self.o.alloc_obs(fields), reset_obs=True)

    def step(self, actions, tape, want_obs):
        obs = self.o.alloc_obs(want_obs) if want_obs else None
        out = self.o.step(actions, tape, obs)
        out["obs"] = obs
        _, _, order, cache = self.o.get_state()
        out["order"], out["cache"] = order, cache
        return out


@pytest.mark.parametrize("name", parity.trace_names())
def test_oracle_matches_reference_trace(name):
    n = parity.replay(name, OracleStepper)
    assert n > 0
