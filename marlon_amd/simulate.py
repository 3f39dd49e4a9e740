"""Batched episode drivers: the counterpart of `marlon.simulate.simulate` -> `marl_algorithm.run_episode`
(marlon/simulate.py:14-35, marlon/baseline_models/multiagent/marl_algorithm.py:144-252) for the step engine.

`run_episode` is the reference loop for a whole batch: per step the attacker predicts and steps, THEN the defender predicts
and steps (marl_algorithm.py:197-240), an env's episode stops as soon as either side reports done (:242-245) or after
`max_steps` loop iterations (:197), and what comes back are the two reward traces.  One episode per env; envs that have
finished are parked (their rows of the traces stay zero).  Rendering (`generate_graph_json`, plotly) is out of scope.

The reference's agents step through a DummyVecEnv, which resets a wrapper as soon as it reports done.  That is visible in
one place: when the ATTACKER ends the episode, its auto-reset notifies the defender wrapper (reset_request, attack_wrapper.py:433-435
-> defend_wrapper.py:479-482), and the defender's step that the loop still takes returns `-1 * last attacker reward` with
truncated=True (defend_wrapper.py:269-271), whatever its action.  `run_episode` returns exactly that value for such envs
(tests/golden/wrap_episode_*.npz, captured from the reference's wrappers in this call order, pin it).

Policies are callables `policy(env) -> actions` (device tensor or array): `attacker_policy(att: AttackerVecEnv)` gives
MultiDiscrete rows [E,10] or Discrete indices [E]; `defender_policy(dfd: DefenderVecEnv)` gives [E,12] rows.
`random_policy` is the counterpart of RandomMarlonAgent / `_step_random_attacker` (marl_algorithm_multi.py:59-76):
uniformly random VALID actions.

`run_episodes` keeps stepping with auto-reset for a fixed number of wrapper steps (throughput-style evaluation of an
attacker); `run_random_agents` is the CyberBattleEnv-level random baseline, entirely on the device.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

from .wrappers import AttackerVecEnv, DefenderVecEnv


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


def random_defender_policy(seed: int = 0) -> Callable[[DefenderVecEnv], object]:
    """Uniform over the defender's MultiDiscrete space [5, N, N, 6, 2, N, 6, 2, N, 3, N, 3] (defend_wrapper.py:162-195)."""
    state = {"gen": None}

    def policy(dfd: DefenderVecEnv):
        t = dfd.torch
        dev = dfd.engine.device
        if state["gen"] is None:
            state["gen"] = t.Generator(device=dev)
            state["gen"].manual_seed(seed)
        nvec = t.as_tensor(dfd.nvec, device=dev, dtype=t.float64)
        return (t.rand((dfd.num_envs, 12), generator=state["gen"], device=dev, dtype=t.float64) * nvec).long()
    return policy


def run_episode(att: AttackerVecEnv, dfd: Optional[DefenderVecEnv] = None, attacker_policy: Optional[Callable] = None,
                defender_policy: Optional[Callable] = None, max_steps: int = 2000) -> Dict[str, object]:
    """marl_algorithm.run_episode for every env of the batch `att` (and of `dfd`, which shares it).  `att` must have been created
    with auto_reset=False: the episode boundary is this loop's business, as in the reference where run_episode stops at the first done.

    Returns device tensors: `attacker_rewards` [T, E] float64 and `defender_rewards` [T, E] float64 (rows past an env's last step are
    zero; `defender_rewards` is None without a defender), `lengths` [E] = entries of the reference's reward lists for that env,
    `attacker_done` / `defender_done` [E] = which side ended it (both False: max_steps reached), and T = loop iterations taken."""
    if att.auto_reset:
        raise ValueError("run_episode needs an AttackerVecEnv created with auto_reset=False (one episode per env, ended by this loop)")
    t = att.torch
    dev = att.engine.device
    E = att.num_envs
    attacker_policy = attacker_policy or random_policy()
    if dfd is not None and defender_policy is None:
        defender_policy = random_defender_policy()
    # attacker_agent.env.reset(); defender_agent.wrapper.on_reset(0); defender_agent.env.reset()   (marl_algorithm.py:176-181)
    att.reset()
    if dfd is not None:
        dfd.reset()
    a_rew, d_rew = [], []
    alive = t.ones(E, dtype=t.bool, device=dev)
    lengths = t.zeros(E, dtype=t.int64, device=dev)
    a_done = t.zeros(E, dtype=t.bool, device=dev)
    d_done = t.zeros(E, dtype=t.bool, device=dev)
    n_steps = 0
    while n_steps < max_steps:
        actions1 = attacker_policy(att)
        _, r1, term1, trunc1, _ = att.step(actions1)
        dones1 = ((term1 | trunc1) != 0) & alive
        r1 = t.where(alive, r1.double(), t.zeros((), dtype=t.float64, device=dev))
        a_rew.append(r1)
        dones2 = t.zeros_like(dones1)
        if dfd is not None:
            actions2 = t.as_tensor(defender_policy(dfd), device=dev).long().clone()
            # envs whose episode is over — earlier, or just now on the attacker's side — take no defender turn on the device:
            # for the latter the reference's defender steps a freshly re-initialised env and its result is discarded in favour of
            # -1 * last attacker reward, truncated (defend_wrapper.py:269-271)
            actions2[~alive | dones1, 0] = -2
            _, r2, term2, trunc2, _ = dfd.step(actions2)
            stepped = alive & ~dones1
            dones2 = (((term2 | trunc2) != 0) & stepped) | dones1
            r2 = t.where(stepped, r2, t.where(dones1, -r1, t.zeros((), dtype=t.float64, device=dev)))
            d_rew.append(r2)
        lengths += alive
        ended = dones1 | dones2
        a_done |= dones1
        d_done |= dones2 & alive
        alive = alive & ~ended
        n_steps += 1
        if not bool(alive.any()):                      # every env's episode is over (one host sync per step; the policies are host-driven anyway)
            break
    out = dict(attacker_rewards=t.stack(a_rew), defender_rewards=t.stack(d_rew) if dfd is not None else None, lengths=lengths,
               attacker_done=a_done, defender_done=d_done, steps=n_steps)
    return out


def uniform_attacker_policy(seed: int = 0) -> Callable[[AttackerVecEnv], object]:
    """`action_space.sample()` of AttackerEnvWrapper's MultiDiscrete space — what RandomMarlonAgent does when the wrapper exposes no
    action masks (random_marlon_agent.py:72-96), i.e. in `marlon.simulate.simulate`'s default universe.  Mostly out-of-range actions."""
    state = {"gen": None}

    def policy(env: AttackerVecEnv):
        t = env.torch
        dev = env.engine.device
        if state["gen"] is None:
            state["gen"] = t.Generator(device=dev)
            state["gen"].manual_seed(seed)
        nvec = t.as_tensor(env.nvec, device=dev, dtype=t.float64)
        return (t.rand((env.num_envs, 10), generator=state["gen"], device=dev, dtype=t.float64) * nvec).long()
    return policy


def simulate(timesteps: int, attacker_option: str = "Random", defender_option: str = "None", attacker_file=None, defender_file=None,
             n_envs: int = 1, seed: int = 0, device: Optional[str] = None, maximum_node_count: int = 100, maximum_total_credentials: int = 1000,
             maximum_discoverable_credentials_per_action: int = 5, attacker_action_masking: bool = False) -> Dict[str, object]:
    """`marlon.simulate.simulate(timesteps, attacker_option, defender_option, attacker_file, defender_file)` (marlon/simulate.py:14-35) for a
    batch of `n_envs` universes: `MultiAgentUniverse.build` (multiagent_universe.py:77-199) with its defaults — CyberBattleToyCtf-v0
    (own_atleast 6, eviction goal), `max_timesteps=2000`, both invalid-action reward modifiers 0 as simulate passes them, and with a
    defender `DefenderConstraint(maintain_sla=0.60)`, `losing_reward=-5000` — then one `run_episode(max_steps=timesteps)`.  Options:
    'Random' (RandomMarlonAgent: uniform over the action masks if the wrapper has them, else uniform over the action space) and 'None'
    (defender only).  'Load' needs Stable-Baselines3 / the Q-learning pickles of the reference and is not available here.
    Returns run_episode's reward traces instead of the reference's plotly frames (rendering is out of scope)."""
    from .cyberbattle_env import AttackerGoal, DefenderConstraint
    from .samples import toy_ctf
    if attacker_option == "None":
        raise ValueError("Attacker cannot be none")
    for opt in (attacker_option, defender_option):
        if opt == "Load":
            raise NotImplementedError("'Load' restores Stable-Baselines3 / Q-compatibility agents; neither library is part of this build")
        if opt not in ("Random", "None"):
            raise ValueError(f"unknown agent option {opt!r}")
    with_defender = defender_option != "None"
    kw = dict(maximum_node_count=maximum_node_count, maximum_total_credentials=maximum_total_credentials,
              maximum_discoverable_credentials_per_action=maximum_discoverable_credentials_per_action,
              attacker_goal=AttackerGoal(own_atleast=6), max_timesteps=2000, invalid_action_reward_modifier=0,
              discrete=attacker_action_masking, auto_reset=False, device=device, seed=seed)
    if with_defender:
        kw.update(defender_constraint=DefenderConstraint(maintain_sla=0.60), losing_reward=-5000.0, learned_defender=True)
    att = AttackerVecEnv(toy_ctf.new_environment(), n_envs, **kw)
    dfd = DefenderVecEnv(att, max_timesteps=2000, invalid_action_reward=0, reset_on_constraint_broken=True, loss_reward=-5000.0) if with_defender else None
    a_pol = random_policy(seed) if attacker_action_masking else uniform_attacker_policy(seed)
    try:
        return run_episode(att, dfd, a_pol, random_defender_policy(seed + 1) if with_defender else None, max_steps=timesteps)
    finally:
        att.close()


def run_episodes(env: AttackerVecEnv, policy: Optional[Callable] = None, max_steps: int = 2000, record_actions: bool = False) -> Dict[str, object]:
    """Step every env of `env` (created with discrete=True for `random_policy`, auto_reset=True) for `max_steps` wrapper steps.
    Returns device tensors: `rewards` [max_steps, E], `dones` [max_steps, E], `episodes` [E] finished episode counts,
    `returns` [E] sum of rewards of finished episodes (and `actions` [max_steps, E, ...] with record_actions)."""
    t = env.torch
    policy = policy or random_policy()
    E = env.num_envs
    rewards = t.zeros((max_steps, E), dtype=t.float32, device=env.engine.device)
    dones = t.zeros((max_steps, E), dtype=t.uint8, device=env.engine.device)
    episodes = t.zeros(E, dtype=t.int64, device=env.engine.device)
    returns = t.zeros(E, dtype=t.float64, device=env.engine.device)
    acts = []
    env.reset()
    for s in range(max_steps):
        a = policy(env)
        if record_actions:
            acts.append(t.as_tensor(a, device=env.engine.device).clone())
        _, r, term, trunc, info = env.step(a)
        d = (term | trunc) != 0
        rewards[s] = r
        dones[s] = d
        episodes += d
        returns += t.where(d, info["episode_return"], t.zeros_like(info["episode_return"]))
    out = dict(rewards=rewards, dones=dones, episodes=episodes, returns=returns)
    if record_actions:
        out["actions"] = t.stack(acts)
    return out


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
