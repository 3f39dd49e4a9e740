"""The Stable-Baselines3 `VecEnv` calling surface over the batched wrappers.

The reference hands its wrappers to SB3 as `VecMonitor(DummyVecEnv([lambda: wrapper, ...]))`
(marlon/baseline_models/ppo_multi/train_marl_multi.py:181-183) or `Monitor(wrapper)` inside a DummyVecEnv SB3 builds itself
(marlon/baseline_models/multiagent/baseline_marlon_agent.py:34,47), and then calls, per rollout step
(baseline_marlon_agent.py:100-167, sb3_contrib `get_action_masks`):

    masks   = np.stack(env.env_method("action_masks"))            # MaskablePPO only
    result  = env.step(actions)                                    # = step_async + step_wait
    new_obs, rewards, dones, infos = result                        # 4-tuple: VecEnv API
    model._update_info_buffer(infos)                               # reads infos[i].get("episode") -> {"r", "l"}
    # on done: infos[i]["terminal_observation"], infos[i]["TimeLimit.truncated"] (collect_rollouts bootstrapping)

`MarlonVecEnv` offers exactly that surface over an `AttackerVecEnv` (MultiDiscrete(10) or Discrete actions) living on the
GPU: DummyVecEnv's auto-reset (the returned observation of an env that ended is its reset observation, the last one of the
episode travels in `infos[i]["terminal_observation"]`), `TimeLimit.truncated = truncated and not terminated`, VecMonitor's
`infos[i]["episode"] = {"r": return, "l": length, "t": seconds}` — the per-env return / length bookkeeping is the fused
`mcbs_attacker_wrapper_post` launch, not a host loop — plus `env_method("action_masks")`, `get_attr`, `set_attr`,
`env_is_wrapped`, `seed`, `close`.  `DefenderVecEnvAdapter` does the same for the learned defender's wrapper.

Outputs are NumPy arrays by default because that is what SB3's buffers take; `numpy_outputs=False` keeps every array a device
tensor (zero copy) for a trainer that consumes torch-ROCm tensors directly.  gymnasium spaces are attached only when
gymnasium is importable (it is not in this image); `nvec` / `discrete_n` describe the action space either way.  Stable-Baselines3
itself is not importable here, so an end-to-end MaskablePPO run is parity-unpinned; what IS pinned (tests/test_gpu_vecenv.py) is
this calling sequence against traces of the reference's own wrapper classes.
"""
from __future__ import annotations

import time
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from .wrappers import AttackerVecEnv, DefenderVecEnv


def _to_numpy(x):
    return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)


class _InfoList(list):
    """infos as SB3 expects them: a list of per-env dicts."""


class MarlonVecEnv:
    """SB3-VecEnv-shaped adapter over `AttackerVecEnv` (which must have been created with auto_reset=True)."""

    metadata = {"render_modes": []}
    render_mode = None

    def __init__(self, venv: AttackerVecEnv, monitor: bool = True, numpy_outputs: bool = True):
        if not venv.auto_reset:
            raise ValueError("MarlonVecEnv needs an AttackerVecEnv with auto_reset=True (DummyVecEnv resets an env that ended inside step_wait)")
        self.venv = venv
        self.num_envs = venv.num_envs
        self.monitor = bool(monitor)
        self.numpy_outputs = bool(numpy_outputs)
        self._actions = None
        self._t_start = time.time()
        self.reset_infos: List[Dict[str, Any]] = [{} for _ in range(self.num_envs)]
        self.observation_space, self.action_space = self._spaces()

    # -- spaces (only with gymnasium; attack_wrapper.py:164-227, action_masking.py:74-80) --
    def _spaces(self):
        try:
            from gymnasium import spaces          # noqa: WPS433 (optional dependency, absent in this image)
        except Exception:
            return None, None
        if not (hasattr(self.venv, "topo") and hasattr(self.venv, "spec")):     # not one of this package's batched wrappers
            return None, None
        v, topo = self.venv, self.venv.topo
        N, Cm, K = v.spec.maximum_node_count, v.spec.maximum_total_credentials, v.spec.maximum_discoverable_credentials_per_action
        L, R, P, NP = len(topo.local_vulnerabilities), len(topo.remote_vulnerabilities), len(topo.ports), len(topo.properties)
        big = np.iinfo(np.int32).max
        obs = {
            "newly_discovered_nodes_count": spaces.Discrete(N + 1), "lateral_move": spaces.Discrete(2), "customer_data_found": spaces.Discrete(2),
            "probe_result": spaces.Discrete(3), "escalation": spaces.Discrete(4), "credential_cache_length": spaces.Discrete(big),
            "discovered_node_count": spaces.Discrete(big),
            "leaked_credentials": spaces.MultiDiscrete(np.tile(np.array([2, Cm, N, P], dtype=np.int32), K)),
            "credential_cache_matrix": spaces.MultiDiscrete(np.tile(np.array([N, P], dtype=np.int32), Cm)),
            "discovered_nodes_properties": spaces.MultiDiscrete(np.full(N * NP, 3, dtype=np.int32)),
            "nodes_privilegelevel": spaces.MultiDiscrete(np.full(N, 4, dtype=np.int32)),
            "local_vulnerability": spaces.MultiBinary([N, L]), "remote_vulnerability": spaces.MultiBinary([N, N, R]),
            "connect": spaces.MultiBinary([N, N, P, Cm]),
        }
        act = spaces.Discrete(v.discrete_n) if v.discrete else spaces.MultiDiscrete(v.nvec)
        return spaces.Dict(obs), act

    # -- conversions --
    def _out(self, x):
        return _to_numpy(x) if self.numpy_outputs else x

    def _obs_out(self, obs: Dict[str, Any]) -> Dict[str, Any]:
        return {k: self._out(v) for k, v in obs.items()}

    # -- VecEnv API --
    def reset(self):
        self._t_start = time.time()
        self.reset_infos = [{} for _ in range(self.num_envs)]
        return self._obs_out(self.venv.reset())

    def step_async(self, actions) -> None:
        self._actions = actions

    def step_wait(self):
        if self._actions is None:
            raise RuntimeError("step_wait() without step_async()")
        actions, self._actions = self._actions, None
        v = self.venv
        obs, rewards, terminated, truncated, info = v.step(actions)
        term, trunc = _to_numpy(terminated).astype(bool), _to_numpy(truncated).astype(bool)
        dones = term | trunc
        invalid = _to_numpy(info["invalid_action"]).astype(bool)
        avail, steps = _to_numpy(info["network_availability"]), _to_numpy(info["step_count"])
        # DummyVecEnv.step_wait: info["TimeLimit.truncated"] = truncated and not terminated
        # (plain Python lists first: one dict per env is what SB3 wants, but per-element NumPy conversions would dominate at 65 536 envs)
        infos = _InfoList({"invalid_action": iv, "cyber_step_executed": not iv, "network_availability": av, "step_count": sc, "TimeLimit.truncated": tl}
                          for iv, av, sc, tl in zip(invalid.tolist(), avail.tolist(), steps.tolist(), (trunc & ~term).tolist()))
        ended = np.flatnonzero(dones)
        if ended.size:
            sel = v.torch.as_tensor(ended, device=v.engine.device)
            term_obs = {k: _to_numpy(x.index_select(0, sel)) for k, x in v.terminal_observation.items()}
            ret, length = _to_numpy(info["episode_return"]), _to_numpy(info["episode_length"])
            elapsed = round(time.time() - self._t_start, 6)
            for j, i in enumerate(ended):
                infos[i]["terminal_observation"] = {k: x[j] for k, x in term_obs.items()}
                if self.monitor:      # VecMonitor.step_wait: episode return / length / wall time of the episode that just ended
                    infos[i]["episode"] = {"r": float(ret[i]), "l": int(length[i]), "t": elapsed}
        rew = _to_numpy(rewards).astype(np.float32) if self.numpy_outputs else rewards
        return self._obs_out(obs), rew, (dones if self.numpy_outputs else (terminated | truncated) != 0), infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def _indices(self, indices) -> Sequence[int]:
        if indices is None:
            return range(self.num_envs)
        if isinstance(indices, int):
            return [indices]
        return list(indices)

    def env_method(self, method_name: str, *method_args, indices=None, **method_kwargs) -> List[Any]:
        """`env_method("action_masks")` is what sb3_contrib's get_action_masks calls: one bool vector per env, np.stack-able."""
        idx = self._indices(indices)
        if method_name == "action_masks":
            m = self.venv.action_masks()
            if indices is not None:
                m = m[list(idx)]
            m = self._out(m)
            return list(m)                     # rows of one [E, A] array: np.stack() of it is a plain copy
        fn = getattr(self.venv, method_name)
        res = fn(*method_args, **method_kwargs)
        return [res for _ in idx]

    def action_masks(self):
        """The [num_envs, A] mask array in one piece (what np.stack(env_method("action_masks")) yields, without the Python list)."""
        return self._out(self.venv.action_masks())

    def get_attr(self, attr_name: str, indices=None) -> List[Any]:
        val = getattr(self.venv, attr_name)
        idx = self._indices(indices)
        if hasattr(val, "shape") and len(getattr(val, "shape", ())) >= 1 and val.shape[0] == self.num_envs:
            arr = _to_numpy(val)
            return [arr[i].item() if arr[i].ndim == 0 else arr[i] for i in idx]
        return [val for _ in idx]

    def set_attr(self, attr_name: str, value, indices=None) -> None:
        if indices is not None and len(self._indices(indices)) != self.num_envs:
            raise ValueError("the batched wrapper holds one value per attribute: set_attr applies to every env")
        setattr(self.venv, attr_name, value)

    def env_is_wrapped(self, wrapper_class, indices=None) -> List[bool]:
        return [False for _ in self._indices(indices)]

    def seed(self, seed: Optional[int] = None) -> List[Optional[int]]:
        return [None if seed is None else seed + i for i in range(self.num_envs)]     # the attacker's step has no randomness to seed

    def get_images(self):
        return [None for _ in range(self.num_envs)]

    def render(self, mode: Optional[str] = None):
        return None

    def close(self) -> None:
        self.venv.close()

    @property
    def unwrapped(self):
        return self


class DefenderVecEnvAdapter:
    """The same VecEnv surface for `DefenderVecEnv` (DefenderEnvWrapper + LearningDefender, defend_wrapper.py:197-327).  The defender
    shares the attacker's environment batch: resets of the environment itself are the attacker side's (MultiAgentUniverse wiring),
    so an episode end here only clears the defender wrapper's own counters for the envs that ended."""

    def __init__(self, dfd: DefenderVecEnv, numpy_outputs: bool = True):
        self.venv = dfd
        self.num_envs = dfd.num_envs
        self.numpy_outputs = bool(numpy_outputs)
        self._actions = None
        self._t_start = time.time()
        t = dfd.torch
        self._ret = t.zeros(self.num_envs, dtype=t.float64, device=dfd.engine.device)
        self._len = t.zeros(self.num_envs, dtype=t.int64, device=dfd.engine.device)
        self.observation_space = self.action_space = None
        try:
            from gymnasium import spaces
            N, S = dfd.attacker.topo.n_nodes, int(dfd.attacker.topo.header()["n_services"])
            self.observation_space = spaces.Dict({"infected_nodes": spaces.MultiBinary(N), "incoming_firewall_status": spaces.MultiBinary(6 * N),
                                                  "outgoing_firewall_status": spaces.MultiBinary(6 * N), "services_status": spaces.MultiBinary(S)})
            self.action_space = spaces.MultiDiscrete(dfd.nvec)
        except Exception:
            pass

    def _out(self, x):
        return _to_numpy(x) if self.numpy_outputs else x

    def reset(self):
        self._t_start = time.time()
        self._ret.zero_()
        self._len.zero_()
        return {k: self._out(v) for k, v in self.venv.reset().items()}

    def step_async(self, actions) -> None:
        self._actions = actions

    def step_wait(self):
        actions, self._actions = self._actions, None
        d = self.venv
        obs, reward, terminated, truncated, info = d.step(actions)
        self._ret += reward
        self._len += 1
        done_t = (terminated | truncated) != 0
        term, trunc = _to_numpy(terminated).astype(bool), _to_numpy(truncated).astype(bool)
        dones = term | trunc
        valid, avail = _to_numpy(info["valid_action"]), _to_numpy(info["network_availability"])
        infos = _InfoList({"valid_action": bool(va), "network_availability": av, "TimeLimit.truncated": tl}
                          for va, av, tl in zip(valid.tolist(), avail.tolist(), (trunc & ~term).tolist()))
        ended = np.flatnonzero(dones)
        if ended.size:
            ret, length = _to_numpy(self._ret), _to_numpy(self._len)
            term_obs = {k: _to_numpy(x)[ended].copy() for k, x in obs.items()}
            elapsed = round(time.time() - self._t_start, 6)
            for j, i in enumerate(ended):
                infos[i]["terminal_observation"] = {k: x[j] for k, x in term_obs.items()}
                infos[i]["episode"] = {"r": float(ret[i]), "l": int(length[i]), "t": elapsed}
            keep = ~done_t
            self._ret *= keep
            self._len *= keep
            obs = d.reset(env_mask=done_t)
        rew = _to_numpy(reward).astype(np.float32) if self.numpy_outputs else reward
        return {k: self._out(v) for k, v in obs.items()}, rew, (dones if self.numpy_outputs else done_t), infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def env_method(self, method_name: str, *args, indices=None, **kwargs) -> List[Any]:
        res = getattr(self.venv, method_name)(*args, **kwargs)
        return [res for _ in (range(self.num_envs) if indices is None else ([indices] if isinstance(indices, int) else indices))]

    def get_attr(self, attr_name: str, indices=None) -> List[Any]:
        val = getattr(self.venv, attr_name)
        idx = range(self.num_envs) if indices is None else ([indices] if isinstance(indices, int) else list(indices))
        if hasattr(val, "shape") and len(val.shape) >= 1 and val.shape[0] == self.num_envs:
            arr = _to_numpy(val)
            return [arr[i].item() if arr[i].ndim == 0 else arr[i] for i in idx]
        return [val for _ in idx]

    def env_is_wrapped(self, wrapper_class, indices=None) -> List[bool]:
        return [False] * (self.num_envs if indices is None else len([indices] if isinstance(indices, int) else indices))

    def seed(self, seed: Optional[int] = None):
        return [None] * self.num_envs

    def close(self) -> None:
        pass        # the environment batch belongs to the attacker side
