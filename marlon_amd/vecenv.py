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
import weakref
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from .wrappers import AttackerVecEnv, DefenderVecEnv


def _to_numpy(x):
    return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)


class _InfoList(list):
    """`infos` as SB3 consumes them — a sequence of one dict per env (`for info in infos: info.get("episode")`,
    `infos[i]["terminal_observation"]`, baseline_marlon_agent.py:100-167, VecMonitor's `infos[:]`) — built ON DEMAND.

    The step leaves the per-env info columns (availability, step count, flags, episode return / length) in ONE host block that one
    asynchronous device-to-host copy fills; nothing is waited for and no Python object per env exists until somebody indexes or iterates
    the list (one dict per env costs ~120 ns: 8 ms at 65 536 envs).  A trainer that consumes arrays reads
    `columns()` / `ended()` instead and never builds a dict.  Envs whose episode ended carry `terminal_observation` and (with a monitor)
    `episode` = {"r", "l", "t"}.  The list must be read before the NEXT step of its env (as SB3 does); if it is still alive and unread
    when that step starts, the adapter loads it first."""

    def __init__(self, n: int, block, ready, layout, term_obs_source, monitor: bool, elapsed: float, keys):
        super().__init__()           # a real `list` (what VecEnv.step_wait is typed to return) whose storage is filled on first use
        self._n, self._block, self._ready, self._layout = n, block, ready, layout
        self._term_src, self._monitor, self._elapsed, self._keys = term_obs_source, monitor, elapsed, keys
        self._cols: Optional[Dict[str, np.ndarray]] = None
        self._ended: Optional[np.ndarray] = None
        self._term_obs: Optional[Dict[str, np.ndarray]] = None
        self._rank: Optional[Dict[int, int]] = None
        self._dicts: Optional[List[Dict[str, Any]]] = None
        self._some: Dict[int, Dict[str, Any]] = {}

    # -- arrays (no per-env objects) --
    def _load(self) -> None:
        if self._cols is not None:
            return
        if self._ready is not None:
            self._ready.synchronize()                       # the one D2H copy of this step's info block
        raw = self._block.numpy() if hasattr(self._block, "numpy") else self._block
        n, cols, off = self._n, {}, 0
        for name, dt in self._layout:
            w = np.dtype(dt).itemsize
            cols[name] = raw[off:off + w * n].view(dt).copy()
            if name in ("invalid_action", "valid_action"):
                cols[name] = cols[name].astype(bool)
            off += w * n
        self._cols = cols
        self._ended = np.flatnonzero((cols["terminated"] | cols["truncated"]) != 0)
        if self._ended.size and self._term_src is not None:
            self._term_obs = self._term_src(self._ended)    # rows of the envs that ended only
        self._rank = {int(i): j for j, i in enumerate(self._ended)}
        self._block = self._ready = self._term_src = None   # the host block goes back to the pinned-memory cache

    def columns(self) -> Dict[str, np.ndarray]:
        """The info fields as NumPy columns of num_envs entries each."""
        self._load()
        return self._cols

    def ended(self) -> np.ndarray:
        """Indices of the envs whose episode ended in this step (terminated or truncated)."""
        self._load()
        return self._ended

    def terminal_observations(self) -> Optional[Dict[str, np.ndarray]]:
        """The last observation of the episodes that ended, rows in the order of ended()."""
        self._load()
        return self._term_obs

    # -- one dict per env --
    def _make(self, i: int) -> Dict[str, Any]:
        c = self._cols
        d = {k: c[src][i].item() if not neg else not c[src][i].item() for k, src, neg in self._keys}
        d["TimeLimit.truncated"] = bool(c["truncated"][i]) and not bool(c["terminated"][i])        # DummyVecEnv.step_wait
        self._finish(d, i)
        return d

    def _finish(self, d: Dict[str, Any], i: int) -> None:
        j = self._rank.get(i)
        if j is None:
            return
        if self._term_obs is not None:
            d["terminal_observation"] = {k: x[j] for k, x in self._term_obs.items()}
        if self._monitor:          # VecMonitor.step_wait: episode return / length / wall time of the episode that just ended
            d["episode"] = {"r": float(self._cols["episode_return"][i]), "l": int(self._cols["episode_length"][i]), "t": self._elapsed}

    def _all(self) -> List[Dict[str, Any]]:
        if self._dicts is None:
            self._load()
            c = self._cols
            # plain Python lists first (per-element NumPy conversions would dominate at 65 536 envs), then one dict literal per env
            tl = ((c["truncated"] != 0) & (c["terminated"] == 0)).tolist()
            if self._keys is MarlonVecEnv._INFO_KEYS:
                ds = [{"invalid_action": iv, "cyber_step_executed": not iv, "network_availability": av, "step_count": sc, "TimeLimit.truncated": t}
                      for iv, av, sc, t in zip(c["invalid_action"].tolist(), c["network_availability"].tolist(), c["step_count"].tolist(), tl)]
            else:
                lists = [([not x for x in c[src].tolist()] if neg else c[src].tolist()) for _, src, neg in self._keys]
                names = [k for k, _, _ in self._keys]
                ds = [dict(zip(names, vals), **{"TimeLimit.truncated": t}) for vals, t in zip(zip(*lists), tl)]
            for i in self._ended.tolist():
                self._finish(ds[i], i)
            for i, d in self._some.items():                # dicts handed out (and possibly edited) before the bulk build stay the ones in the list
                ds[i] = d
            list.extend(self, ds)                          # from here on this IS an ordinary list of dicts
            self._dicts = self
        return self

    def _filled(name):                                     # every other list method: fill the storage first, then behave like a list
        def method(self, *args, **kwargs):
            self._all()
            return getattr(list, name)(self, *args, **kwargs)
        method.__name__ = name
        return method

    for _name in ("__contains__", "__reversed__", "__add__", "__iadd__", "__mul__", "__imul__", "__delitem__", "__lt__", "__le__", "__gt__", "__ge__",
                  "__ne__", "append", "extend", "insert", "pop", "remove", "clear", "index", "count", "sort", "reverse", "copy", "__reduce_ex__"):
        locals()[_name] = _filled(_name)
    del _name, _filled

    def __len__(self) -> int:
        return self._n

    def __iter__(self):
        self._all()
        return list.__iter__(self)

    def __getitem__(self, i):
        if isinstance(i, slice) or self._dicts is not None:
            self._all()
            return list.__getitem__(self, i)
        i = int(i)
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError("info index out of range")
        d = self._some.get(i)
        if d is None:
            self._load()
            d = self._some[i] = self._make(i)
        return d

    def __setitem__(self, i, value) -> None:
        self._all()
        list.__setitem__(self, i, value)

    def __eq__(self, other) -> bool:
        self._all()
        return list.__eq__(self, other)

    __hash__ = None

    def __repr__(self) -> str:
        return f"<infos of {self._n} envs, {'built' if self._dicts is not None else 'not built yet'}>"


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
        self._last_infos = None
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

    _INFO_LAYOUT = (("network_availability", "<f8"), ("episode_return", "<f8"), ("rewards", "<f4"), ("step_count", "<i4"), ("episode_length", "<i4"),
                    ("invalid_action", "u1"), ("terminated", "u1"), ("truncated", "u1"))
    # info dict key -> (column, negated)
    _INFO_KEYS = (("invalid_action", "invalid_action", False), ("cyber_step_executed", "invalid_action", True),
                  ("network_availability", "network_availability", False), ("step_count", "step_count", False))

    def _info_block(self, rewards, terminated, truncated, info):
        """The step's per-env info columns gathered into ONE device buffer (one launch) and sent to pinned host memory by ONE asynchronous
        copy; an event marks its arrival.  Replaces six pageable device-to-host copies per step."""
        t = self.venv.torch
        u8 = t.uint8
        src = {"network_availability": info["network_availability"], "episode_return": info["episode_return"], "rewards": rewards,
               "step_count": info["step_count"], "episode_length": info["episode_length"], "invalid_action": info["invalid_action"],
               "terminated": terminated, "truncated": truncated}
        dev = t.cat([src[name].contiguous().view(u8).reshape(-1) for name, _ in self._INFO_LAYOUT])
        if not dev.is_cuda:          # (host tensors: the CPU stand-ins of tests/test_host_logic.py)
            return dev, None
        host = t.empty(dev.shape, dtype=u8, pin_memory=True)
        host.copy_(dev, non_blocking=True)
        ready = t.cuda.Event()
        ready.record()
        return host, ready

    def _terminal_rows(self, ended: np.ndarray) -> Dict[str, np.ndarray]:
        v = self.venv
        sel = v.torch.as_tensor(ended, device=v.engine.device)
        return {k: _to_numpy(x.index_select(0, sel)) for k, x in v.terminal_observation.items()}

    def step_wait(self):
        if self._actions is None:
            raise RuntimeError("step_wait() without step_async()")
        actions, self._actions = self._actions, None
        v = self.venv
        last = self._last_infos() if self._last_infos is not None else None
        if last is not None:
            last._load()             # an unread info list of the previous step reads the wrapper's buffers before this step overwrites them
        obs, rewards, terminated, truncated, info = v.step(actions)
        host, ready = self._info_block(rewards, terminated, truncated, info)
        infos = _InfoList(self.num_envs, host, ready, self._INFO_LAYOUT, self._terminal_rows, self.monitor, round(time.time() - self._t_start, 6),
                          self._INFO_KEYS)
        self._last_infos = weakref.ref(infos)
        if not self.numpy_outputs:   # device tensors out, nothing waited for: the trainer consumes tensors and reads `infos` only if it wants to
            return self._obs_out(obs), rewards, (terminated | truncated) != 0, infos
        cols = infos.columns()       # (waits for the block: SB3 takes NumPy rewards / dones)
        dones = (cols["terminated"] | cols["truncated"]) != 0
        return self._obs_out(obs), cols["rewards"].astype(np.float32), dones, infos

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
        infos = [{"valid_action": bool(va), "network_availability": av, "TimeLimit.truncated": tl}
                 for va, av, tl in zip(valid.tolist(), avail.tolist(), (trunc & ~term).tolist())]
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
