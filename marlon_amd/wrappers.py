"""Batched counterparts of marlon's attacker env wrappers.

`AttackerVecEnv` is `AttackerEnvWrapper` (marlon/baseline_models/env_wrappers/attack_wrapper.py) for a whole batch
of environments living on the GPU, with the Stable-Baselines3 `VecEnv` calling convention the reference reaches
through `DummyVecEnv([...])` (marlon/baseline_models/ppo_multi/train_marl_multi.py:181-183):

  * action: MultiDiscrete `[3, N, L, N, N, R, N, N, P, C]` rows `[kind, l_src, l_vuln, r_src, r_tgt, r_vuln, c_src,
    c_tgt, c_port, c_cred]` (attack_wrapper.py:206-227) or, with `discrete=True`, the single Discrete index of
    `MaskedDiscreteAttackerWrapper` (action_masking.py:30-142) — decoded on the device by `mcbs_decode_attacker_actions`;
  * an action whose node index is not discovered yet does NOT step the env: the last observation is returned,
    reward is `0 + invalid_action_reward_modifier`, `info["invalid_action"]` is set (attack_wrapper.py:286-308);
  * observation: the flat dict of attack_wrapper.py:474-522, every value a device tensor with a leading env axis
    (`scalars` also split into the seven named integer keys); `action_masks()` = action_masking.py:90-110;
  * `timesteps` counts wrapper steps (invalid ones included) and truncates at `max_timesteps` (:346-352);
  * finished envs are reset inside `step` like a VecEnv does; the observation that ended the episode is kept in
    `terminal_observation` for the envs flagged in `dones`;
  * a step's output tensors (rewards, flags, info) are valid until the step after next (two alternating sets; with `use_graph` the
    one persistent set is overwritten by the next step); the observation tensors are the wrapper's own persistent buffers.

All simulation work is in the HIP kernels; the few tensor expressions here are the wrapper's own bookkeeping
(timestep counters, reward modifier).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from ._abi import EnvSpec
from .cyberbattle_env import (SCALAR_KEYS, AttackerGoal, DefenderConstraint, DefenderGoal, spec_from_kwargs)
from .flatten import FlatTopology, flatten

# what the engine writes; the three separate action masks of marlon's observation dict are VIEWS into mask_discrete
# (connect | local | remote, action_masking.py:96-110): the same bytes, written once
FLAT_FIELDS = ["scalars", "leaked_credentials", "credential_cache_matrix", "discovered_nodes_properties",
               "nodes_privilegelevel", "mask_discrete"]


class AttackerVecEnv:
    def __init__(self, initial_environment, n_envs: int, maximum_total_credentials: int = 1000, maximum_node_count: int = 100,
                 maximum_discoverable_credentials_per_action: int = 5, defender_agent=None,
                 attacker_goal: Optional[AttackerGoal] = AttackerGoal(own_atleast_percent=1.0),
                 defender_goal=DefenderGoal(eviction=True), defender_constraint=DefenderConstraint(maintain_sla=0.0),
                 winning_reward=5000.0, losing_reward=0.0, max_timesteps: int = 2000, invalid_action_reward_modifier=-1,
                 discrete: bool = False, auto_reset: bool = True, device: Optional[str] = None, seed: int = 0,
                 env_id_base: int = 0, rng_kind: int = 0, learned_defender: bool = False, materialize_masks: bool = True,
                 use_graph: bool = False):
        from .engine import BatchEngine
        self.topo: FlatTopology = initial_environment if isinstance(initial_environment, FlatTopology) else flatten(initial_environment)
        # the wrapper owns truncation and resets (its clock counts invalid actions too), so the engine's own are off
        self.spec: EnvSpec = spec_from_kwargs(n_envs, maximum_total_credentials, maximum_node_count,
                                              maximum_discoverable_credentials_per_action, defender_agent, attacker_goal,
                                              defender_goal, defender_constraint, winning_reward, losing_reward,
                                              auto_reset=False, max_episode_steps=0, seed=seed, env_id_base=env_id_base, rng_kind=rng_kind)
        if learned_defender:
            if defender_agent is not None:
                raise ValueError("a learned defender replaces the in-env defender_agent (multiagent_universe.py:160-167)")
            self.spec.defender = ("external",)
        self.engine = BatchEngine(self.topo, self.spec, device=device)
        t = self.torch = self.engine.torch
        self.num_envs = n_envs
        self.max_timesteps = int(max_timesteps)
        self.invalid_action_reward_modifier = float(invalid_action_reward_modifier)
        self.discrete = bool(discrete)
        self.auto_reset = bool(auto_reset)
        N, Cm = maximum_node_count, maximum_total_credentials
        L, R, P = len(self.topo.local_vulnerabilities), len(self.topo.remote_vulnerabilities), len(self.topo.ports)
        self.nvec = np.array([3, N, L, N, N, R, N, N, P, Cm], dtype=np.int64)                      # attack_wrapper.py:206-227
        self.discrete_n = N * N * P * Cm + N * L + N * N * R                                         # action_masking.py:74-80
        dev = self.engine.device
        # materialize_masks=False: the three action masks (94 % of the observation's bytes) are not written at all; a policy applies them
        # to its logits with mask_logits() (mcbs_mask_logits: rebuilt on the device from the observation's digest).  The observation
        # dict then has no local_vulnerability / remote_vulnerability / connect entries and action_masks() raises.
        self.materialize_masks = bool(materialize_masks)
        if self.materialize_masks:
            # rows of the flat mask padded to whole 128-byte lines: dense rows (Chain-10: 14 172 bytes) share cache lines with their
            # neighbours, which cost the Discrete observation a quarter of its write bandwidth; consumers see [:, :discrete_n] views
            self.engine.set_mask_discrete_stride((self.discrete_n + 127) // 128 * 128)
        self._obs = self.engine.alloc_obs(FLAT_FIELDS if self.materialize_masks else FLAT_FIELDS[:-1])
        self._terminal = {k: t.zeros_like(v) for k, v in self._obs.items()}
        self._mask_split = (N * N * P * Cm, N * L, (N, N, P, Cm), (N, L), (N, N, R))
        self._rows = t.zeros((n_envs, 5), dtype=t.int32, device=dev)
        self._invalid = t.zeros(n_envs, dtype=t.uint8, device=dev)
        self.timesteps = t.zeros(n_envs, dtype=t.int32, device=dev)
        self.valid_action_count = t.zeros(n_envs, dtype=t.int64, device=dev)
        self.invalid_action_count = t.zeros(n_envs, dtype=t.int64, device=dev)
        self.episode_returns = t.zeros(n_envs, dtype=t.float64, device=dev)
        self.last_cyber_reward = t.zeros(n_envs, dtype=t.float32, device=dev)
        self.has_cyber_reward = t.zeros(n_envs, dtype=t.bool, device=dev)    # AttackerEnvWrapper.cyber_rewards is non-empty
        # outputs of the fused bookkeeping / auto-reset launch (mcbs_attacker_wrapper_finish) and its argument blocks
        self._rewards = t.zeros(n_envs, dtype=t.float32, device=dev)
        self._truncated = t.zeros(n_envs, dtype=t.uint8, device=dev)
        self._dones = t.zeros(n_envs, dtype=t.uint8, device=dev)
        self._ret_out = t.zeros(n_envs, dtype=t.float64, device=dev)
        self._len_out = t.zeros(n_envs, dtype=t.int32, device=dev)
        self._n_done = t.zeros(1, dtype=t.int32, device=dev)
        self._wb = self._keep = self._fresh = self._reset_rows = None
        self._calls = {}                                   # bound library calls (engine.wrapper_step_call), one per output set and settings
        self._dev_index = self.engine.device.index if self.engine.device.index is not None else t.cuda.current_device()
        self._action_shape = (n_envs,) if self.discrete else (n_envs, 10)
        self._obs_views = self._terminal_views = None
        self._ring, self._slot = None, 0
        self._one_launch = None
        # use_graph: the whole wrapper step (decode, environment step + observation, bookkeeping, terminal-observation copy, reset and reset
        # observation of the envs that ended) is captured into ONE hipGraph on the first call and replayed afterwards: the step has no
        # host round trip, so what remains on the host is one graph launch.  Outputs are then the wrapper's own buffers (overwritten
        # by the next step) instead of copies.  Not with a draw tape (its address changes per step).
        self.use_graph = bool(use_graph)
        if self.use_graph and rng_kind == 1:
            raise ValueError("use_graph replays fixed kernel arguments: it cannot be combined with a defender draw tape (rng_kind=TAPE)")
        self._graph = None
        self._act_in = t.zeros((n_envs,) if self.discrete else (n_envs, 10), dtype=t.int64, device=dev)
        self._executed = t.zeros(n_envs, dtype=t.bool, device=dev)
        self._graph_out = self._terminated_out = None
        self.reset()

    # -- observation plumbing --
    def _public(self, obs: Dict[str, object]) -> Dict[str, object]:
        M, ML, s_connect, s_local, s_remote = self._mask_split
        out = {}
        if "mask_discrete" in obs:
            flat = obs["mask_discrete"][:, :self.discrete_n]          # (rows are padded to whole cache lines)
            out = {"local_vulnerability": flat[:, M:M + ML].unflatten(1, s_local), "remote_vulnerability": flat[:, M + ML:].unflatten(1, s_remote),
                   "connect": flat[:, :M].unflatten(1, s_connect)}
        out.update({
               "leaked_credentials": obs["leaked_credentials"].reshape(self.num_envs, -1),
               "credential_cache_matrix": obs["credential_cache_matrix"].reshape(self.num_envs, -1),
               "discovered_nodes_properties": obs["discovered_nodes_properties"].reshape(self.num_envs, -1),
               "nodes_privilegelevel": obs["nodes_privilegelevel"]})
        for i, k in enumerate(SCALAR_KEYS):
            out[k] = obs["scalars"][:, i]
        return out

    @property
    def observation(self) -> Dict[str, object]:
        # views of the wrapper's persistent observation tensors: built once (a dozen view operations cost more host time than the
        # device needs for the whole step)
        if self._obs_views is None:
            self._obs_views = self._public(self._obs)
        return dict(self._obs_views)

    @property
    def terminal_observation(self) -> Dict[str, object]:
        if self._terminal_views is None:
            self._terminal_views = self._public(self._terminal)
        return dict(self._terminal_views)

    def action_masks(self):
        """[n_envs, N*N*P*C + N*L + N*N*R] bool, MaskedDiscreteAttackerWrapper order (connect, local, remote)."""
        if not self.materialize_masks:
            raise RuntimeError("this AttackerVecEnv was created with materialize_masks=False: apply the mask with mask_logits(logits)")
        return self._obs["mask_discrete"].view(self.torch.bool)[:, :self.discrete_n]      # the int8 mask holds 0 / 1 only: a bool view (rows padded
                                                                                           # to whole cache lines), no second gigabyte

    def mask_logits(self, logits, fill: float = -1e8):
        """`where(action_masks(), logits, fill)` in place on the device, without the mask: what MaskablePPO's MaskableCategorical does
        with action_masks() (train_marl_multi.py:259-293), straight from the digest of the observation this env last returned."""
        return self.engine.mask_logits(logits, fill)

    # -- VecEnv surface --
    def reset(self):
        self.engine.reset()
        self.engine.observe(self._obs)
        if self._reset_rows is None:
            self._reset_rows = {k: v[0:1].clone() for k, v in self._obs.items()}
        self.timesteps.zero_()
        self.valid_action_count.zero_()
        self.invalid_action_count.zero_()
        self.episode_returns.zero_()
        self.has_cyber_reward.zero_()
        return self.observation

    @property
    def action_buffer(self):
        """use_graph: the int64 device tensor the captured step reads its actions from ([E] Discrete, [E, 10] MultiDiscrete).  A policy
        that writes into it and calls `step(venv.action_buffer)` saves the copy."""
        return self._act_in

    def _step_device(self, actions) -> None:
        """Everything a wrapper step does on the device, enqueued on the current stream by ONE call into the library
        (mcbs_attacker_wrapper_step) without any host round trip: decode + interception of out-of-range actions, environment step and
        observation, then — one launch — counters, reward modifier of intercepted actions (attack_wrapper.py:296,354), truncation
        (:350-352), episode returns and what DummyVecEnv.step_wait does for an env that reports done: keep its last observation, reset
        it, return the reset observation (every env resets to the same state, so that is row 0 of reset()'s observation)."""
        t = self.torch
        a = actions
        if not (type(a) is t.Tensor and a.dtype is t.int64 and a.is_cuda and a.get_device() == self._dev_index and a.is_contiguous()):
            a = a if isinstance(a, t.Tensor) else t.as_tensor(np.asarray(a))       # (the common case — the policy's own int64 tensor — skips this)
            a = a.to(device=self.engine.device, dtype=t.int64).contiguous()
        want = self._action_shape
        if a.shape != want:
            raise ValueError(f"expected actions of shape {want}, got {tuple(a.shape)}")
        if self._keep is None:
            eng = self.engine
            self._keep = eng.row_copies([(self._obs[k], self._terminal[k]) for k in self._obs])
            self._fresh = eng.row_copies([(self._reset_rows[k], self._obs[k]) for k in self._obs], one_row_src=True)
            self._obs_block = eng.obs_struct(self._obs)
        if self._wb is None:
            self._wb = self._wrapper_buffers(self._invalid, self.engine.terminated if self.use_graph else self._terminated_out, self._rewards,
                                             self._truncated, self._ret_out, self._len_out, self._executed)
        # the library call with every argument block bound once per (output set, wrapper settings): engine.wrapper_step_call
        key = (id(self._wb), self.invalid_action_reward_modifier, self.max_timesteps, self.auto_reset)
        call = self._calls.get(key)
        if call is None:
            call = self._calls[key] = self.engine.wrapper_step_call(self.discrete, self._rows, self._obs_block, self._wb, self.invalid_action_reward_modifier,
                                                                    self.max_timesteps, self.auto_reset, self._keep, self._fresh)
        call(a.data_ptr())

    def _wrapper_buffers(self, invalid, terminated, rewards, truncated, ret_out, len_out, executed):
        from ._abi import WrapperBuffers
        return WrapperBuffers(*[x.data_ptr() for x in (
            invalid, self.engine.reward, terminated, self.timesteps, self.valid_action_count, self.invalid_action_count, self.episode_returns,
            self.last_cyber_reward, self.has_cyber_reward, rewards, truncated, self._dones, ret_out, len_out, self._n_done, executed)])

    def step(self, actions):
        """-> (observation dict, rewards f32 [E], terminated u8 [E], truncated u8 [E], info dict of tensors)."""
        t = self.torch
        if self.use_graph:
            if actions is not self._act_in:                    # a policy may write its actions straight into `action_buffer`: no copy then
                a = actions if isinstance(actions, t.Tensor) else t.as_tensor(np.asarray(actions))
                self._act_in.copy_(a.to(device=self.engine.device).reshape(self._act_in.shape), non_blocking=True)
            if self._one_launch is None:
                self._one_launch = self.engine.wrapper_step_launches(self.materialize_masks) == 1
            if self._one_launch:
                # the whole step is ONE kernel: launched directly on the persistent buffers (a one-node hipGraph replay costs the device
                # ~8 us more than the launch it wraps: 30 vs 20 us per step at 65 536 Chain-10 envs)
                self._step_device(self._act_in)
            elif self._graph is None:
                self._step_device(self._act_in)            # this step runs eagerly (it also creates the argument blocks) ...
                t.cuda.synchronize(self.engine.device)
                g = t.cuda.CUDAGraph()
                side = t.cuda.Stream(device=self.engine.device)
                side.wait_stream(t.cuda.current_stream(self.engine.device))
                with t.cuda.stream(side):
                    with t.cuda.graph(g, stream=side):
                        self._step_device(self._act_in)
                t.cuda.current_stream(self.engine.device).wait_stream(side)
                self._graph = g
                # ... and is then captured for the steps to come (capturing executes nothing)
            else:
                self._graph.replay()
            if self._graph_out is None:                        # the same persistent buffers every step: built once
                self._graph_out = (self.observation, self._rewards, self.engine.terminated, self._truncated,
                                   {"invalid_action": self._invalid.view(t.bool), "cyber_step_executed": self._executed,
                                    "network_availability": self.engine.info["network_availability"], "step_count": self.engine.info["step_count"],
                                    "episode_return": self._ret_out, "episode_length": self._len_out})
            return self._graph_out
        # eager: the step's outputs alternate between TWO sets of tensors allocated once (a step's outputs stay valid until the step after
        # next; consumers that keep them longer clone them).  Seven allocations and a rebuilt argument block per step cost more host time
        # than the device needs for the whole wrapper step (20 us at 65 536 envs).
        E, dev = self.num_envs, self.engine.device
        if self._ring is None:
            self._ring = []
            for _ in range(2):
                o = dict(invalid=t.empty(E, dtype=t.uint8, device=dev), executed=t.empty(E, dtype=t.bool, device=dev),
                         rewards=t.empty(E, dtype=t.float32, device=dev), truncated=t.empty(E, dtype=t.uint8, device=dev),
                         ret=t.empty(E, dtype=t.float64, device=dev), length=t.empty(E, dtype=t.int32, device=dev),
                         terminated=t.empty(E, dtype=t.uint8, device=dev))
                wb = self._wrapper_buffers(o["invalid"], o["terminated"], o["rewards"], o["truncated"], o["ret"], o["length"], o["executed"])
                info = {"invalid_action": o["invalid"].view(t.bool), "cyber_step_executed": o["executed"],
                        "network_availability": self.engine.info["network_availability"], "step_count": self.engine.info["step_count"],
                        "episode_return": o["ret"], "episode_length": o["length"]}
                self._ring.append((o, wb, info))
        self._slot ^= 1
        o, wb, info = self._ring[self._slot]
        self._invalid, self._executed, self._rewards, self._truncated = o["invalid"], o["executed"], o["rewards"], o["truncated"]
        self._ret_out, self._len_out, self._terminated_out, self._wb = o["ret"], o["length"], o["terminated"], wb
        self._step_device(actions)
        return self.observation, o["rewards"], o["terminated"], o["truncated"], dict(info)

    def close(self) -> None:
        self.engine.close()


class DefenderVecEnv:
    """`DefenderEnvWrapper` + `LearningDefender` (marlon/baseline_models/env_wrappers/defend_wrapper.py,
    marlon/defender_agents/defender.py) for the batch an `AttackerVecEnv(..., learned_defender=True)` owns: the two
    wrappers share one environment batch, as in MultiAgentUniverse.build (multiagent_universe.py:160-199).

    step(actions[E,12]) = validity check + executeAction AND the wrapper's reward shaping in one launch, then the observation
    (`mcbs_defender_wrapper_step`).  The shaping (defend_wrapper.py:228-282): invalid-action penalty, minus the attacker's last environment reward, a one-time
    `loss_reward` when availability first drops below `maintain_sla` (terminating if reset_on_constraint_broken), a
    worsening penalty while breached, `winning_reward` on eviction; truncation at max_timesteps.
    The learned defender always acts on the live environment."""

    def __init__(self, attacker: AttackerVecEnv, max_timesteps: int = 100, invalid_action_reward: float = 0.0,
                 reset_on_constraint_broken: bool = True, loss_reward: float = -5000.0, sla_worsening_penalty_scale: float = 200.0,
                 use_graph: bool = False):
        self.attacker = attacker
        self.engine = attacker.engine
        t = self.torch = attacker.torch
        E, dev = attacker.num_envs, self.engine.device
        self.num_envs = E
        self.max_timesteps = int(max_timesteps)
        self.invalid_action_penalty = float(invalid_action_reward)
        self.reset_on_constraint_broken = bool(reset_on_constraint_broken)
        self.loss_reward = float(loss_reward)
        self.sla_worsening_penalty_scale = float(sla_worsening_penalty_scale)
        self.maintain_sla = float(attacker.spec.maintain_sla)
        self.winning_reward = float(attacker.spec.winning_reward)
        N = attacker.topo.n_nodes
        self.nvec = np.array([5, N, N, 6, 2, N, 6, 2, N, 3, N, 3], dtype=np.int64)          # defend_wrapper.py:162-195
        self._obs = self.engine.alloc_defender_obs()
        self.timesteps = t.zeros(E, dtype=t.int32, device=dev)
        self.has_breached_sla = t.zeros(E, dtype=t.bool, device=dev)
        self.prev_availability = t.ones(E, dtype=t.float64, device=dev)
        self.valid_action_count = t.zeros(E, dtype=t.int64, device=dev)
        self.invalid_action_count = t.zeros(E, dtype=t.int64, device=dev)
        self._evicted = None
        self._ring, self._slot = None, 0
        # use_graph: outputs are the wrapper's own persistent buffers (overwritten by the next step) and the actions are copied into a
        # persistent buffer; the turn's two launches are issued directly (see step)
        self.use_graph = bool(use_graph)
        self._act_in = t.zeros((E, 12), dtype=t.int64, device=dev)
        self.reset()

    @property
    def observation(self) -> Dict[str, object]:
        return self._obs

    def reset(self, env_mask=None):
        """Wrapper state of the envs in env_mask (all if None); the environment itself is reset by the attacker side."""
        t = self.torch
        self.engine.defender_observe(self._obs)
        avail = self.engine.step_info()["network_availability"]
        keep = t.zeros(self.num_envs, dtype=t.bool, device=self.engine.device) if env_mask is None else ~(env_mask != 0)
        self.timesteps *= keep
        self.valid_action_count *= keep
        self.invalid_action_count *= keep
        self.has_breached_sla &= keep
        self.prev_availability.copy_(t.where(keep, self.prev_availability, avail))    # in place: the fused shaping launch holds its address
        return self._obs

    def step(self, actions):
        """-> (observation dict, reward f64 [E], terminated u8 [E], truncated u8 [E], info)."""
        t = self.torch
        if self.use_graph:
            # (persistent outputs, actions copied into a persistent buffer — what a captured turn offered; the turn itself is two
            # launches issued directly: replaying them from a hipGraph cost the device more than the launches it wraps, 23 vs 12 us
            # per turn at 16 384 ToyCtf envs)
            a = actions if isinstance(actions, t.Tensor) else t.as_tensor(np.asarray(actions))
            self._act_in.copy_(a.to(device=self.engine.device).reshape(self._act_in.shape), non_blocking=True)
            actions = self._act_in
        # eager: ONE library call per turn (mcbs_defender_wrapper_step: the turn and the reward shaping in one launch, then the observation);
        # the turn's outputs alternate between two sets of tensors allocated once (valid until the turn after next)
        E, dev = self.num_envs, self.engine.device
        if self._ring is None:
            from ._abi import DefenderObs, DefenderWrapperBuffers, DefenderWrapperCfg
            self._evicted = t.zeros(E, dtype=t.uint8, device=dev)
            self._obs_block = DefenderObs(**{k: x.data_ptr() for k, x in self._obs.items()})
            self._wc = DefenderWrapperCfg(self.invalid_action_penalty, self.loss_reward, self.sla_worsening_penalty_scale, self.maintain_sla,
                                          self.winning_reward, int(self.reset_on_constraint_broken), self.max_timesteps)
            self._ring = []
            for _ in range(2):
                o = dict(valid=t.empty(E, dtype=t.uint8, device=dev), availability=t.empty(E, dtype=t.float64, device=dev),
                         reward=t.empty(E, dtype=t.float64, device=dev), terminated=t.empty(E, dtype=t.uint8, device=dev),
                         truncated=t.empty(E, dtype=t.uint8, device=dev), breached=t.empty(E, dtype=t.uint8, device=dev),
                         won=t.empty(E, dtype=t.uint8, device=dev))
                wb = DefenderWrapperBuffers(*[x.data_ptr() for x in (
                    o["valid"], o["availability"], self._evicted, self.attacker.has_cyber_reward, self.attacker.last_cyber_reward, self.timesteps,
                    self.valid_action_count, self.invalid_action_count, self.has_breached_sla, self.prev_availability, o["reward"], o["terminated"],
                    o["truncated"], o["breached"], o["won"])])
                info = {"valid_action": o["valid"].view(t.bool), "network_availability": o["availability"], "sla_breached": o["breached"].view(t.bool),
                        "defender_won": o["won"].view(t.bool)}
                self._ring.append((o, wb, info, self.engine.defender_wrapper_step_call(self._obs_block, wb, self._wc)))
        if not self.use_graph:
            self._slot ^= 1
        o, wb, info, call = self._ring[self._slot]
        a = actions
        if not (type(a) is t.Tensor and a.dtype is t.int64 and a.is_cuda and a.get_device() == self.attacker._dev_index and a.is_contiguous()):
            a = a if isinstance(a, t.Tensor) else t.as_tensor(np.asarray(a))
            a = a.to(device=dev, dtype=t.int64).contiguous()
        if a.shape != (E, 12):
            raise ValueError(f"defender actions must have shape ({E}, 12), got {tuple(a.shape)}")
        call(a.data_ptr())
        self._out = o
        return self._obs, o["reward"], o["terminated"], o["truncated"], dict(info)
