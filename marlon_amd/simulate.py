"""Batched episode driver: the counterpart of `marlon.simulate.simulate` -> `marl_algorithm.run_episode`
(marlon/simulate.py:14-35, marlon/baseline_models/multiagent/marl_algorithm.py:144-252) for the step engine.

The reference loop is `attacker.predict -> attacker.env.step -> [defender acts] -> record` until done or
`max_steps`, one env at a time, returning plotly frames.  Here the same loop runs for `n_envs` environments at once
on the device and returns per-env reward traces; the in-env defender (ScanAndReimage) acts inside the step kernel.
Policies are callables `policy(env: AttackerVecEnv) -> actions` (device tensor); `random_policy` is the counterpart
of RandomMarlonAgent / `_step_random_attacker` (marl_algorithm_multi.py:59-76): uniformly random VALID actions.
Rendering (plotly graphs, simulation.py) is out of scope.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

from .wrappers import AttackerVecEnv


def random_policy(seed: int = 0) -> Callable[[AttackerVecEnv], object]:
    """Uniform over the currently valid Discrete actions (what MaskablePPO's masked sampling does at initialisation)."""
    state = {"gen": None}

    def policy(env: AttackerVecEnv):
        t = env.torch
        if state["gen"] is None:
            state["gen"] = t.Generator(device=env.engine.device)
            state["gen"].manual_seed(seed)
        mask = env.action_masks()
        scores = t.rand(mask.shape, generator=state["gen"], device=mask.device)
        return t.where(mask, scores, t.full_like(scores, -1.0)).argmax(dim=1)
    return policy


def run_episodes(env: AttackerVecEnv, policy: Optional[Callable] = None, max_steps: int = 2000) -> Dict[str, object]:
    """Step every env of `env` (created with discrete=True for `random_policy`) for `max_steps` wrapper steps.
    Returns device tensors: `rewards` [max_steps, E], `dones` [max_steps, E], `episodes` [E] finished episode counts,
    `returns` [E] sum of rewards of finished episodes."""
    t = env.torch
    policy = policy or random_policy()
    E = env.num_envs
    rewards = t.zeros((max_steps, E), dtype=t.float32, device=env.engine.device)
    dones = t.zeros((max_steps, E), dtype=t.uint8, device=env.engine.device)
    episodes = t.zeros(E, dtype=t.int64, device=env.engine.device)
    returns = t.zeros(E, dtype=t.float64, device=env.engine.device)
    env.reset()
    for s in range(max_steps):
        _, r, term, trunc, info = env.step(policy(env))
        d = (term | trunc) != 0
        rewards[s] = r
        dones[s] = d
        episodes += d
        returns += t.where(d, info["episode_return"], t.zeros_like(info["episode_return"]))
    return dict(rewards=rewards, dones=dones, episodes=episodes, returns=returns)


def run_random_agents(engine, n_steps: int, valid: bool = True, seed: int = 0, chunk: int = 256) -> Dict[str, object]:
    """Random attackers at the CyberBattleEnv level — the loop `env.step(env.sample_valid_action())` of CyberBattleSim's random
    baseline and of marlon's RandomMarlonAgent — for every env of a BatchEngine, entirely on the device: `mcbs_rollout_random`
    samples each action inside the step kernel, `chunk` steps per launch.  Returns device tensors `rewards` / `dones`
    [n_steps, E] (auto-reset and truncation as configured in the batch's EnvSpec)."""
    t = engine.torch
    rewards = t.empty((n_steps, engine.E), dtype=t.float32, device=engine.device)
    dones = t.empty((n_steps, engine.E), dtype=t.uint8, device=engine.device)
    for s in range(0, n_steps, chunk):
        k = min(chunk, n_steps - s)
        r, d, _ = engine.rollout_random(k, valid=valid, seed=seed, first_step=s)
        rewards[s:s + k], dones[s:s + k] = r, d
    return dict(rewards=rewards, dones=dones)
