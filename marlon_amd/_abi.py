"""ctypes mirrors of the structs in include/mcbs.h (same field order, natural alignment)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

ABI_VERSION = 1
DEFENDER_NONE, DEFENDER_SCAN_AND_REIMAGE, DEFENDER_EXTERNAL, DEFENDER_RANDOM_EVENTS = 0, 1, 2, 3
RNG_PHILOX, RNG_TAPE = 0, 1


class BatchCfg(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("n_envs", C.c_uint32), ("device", C.c_int32),
        ("maximum_node_count", C.c_uint32), ("maximum_total_credentials", C.c_uint32),
        ("maximum_discoverable_credentials_per_action", C.c_uint32),
        ("has_attacker_goal", C.c_uint32), ("goal_own_atleast", C.c_uint32),
        ("goal_reward", C.c_double), ("goal_low_availability", C.c_double), ("goal_own_atleast_percent", C.c_double),
        ("defender_goal_eviction", C.c_uint32), ("defender_kind", C.c_uint32),
        ("maintain_sla", C.c_double), ("winning_reward", C.c_double), ("losing_reward", C.c_double),
        ("scan_probability", C.c_double), ("scan_capacity", C.c_uint32), ("scan_frequency", C.c_uint32),
        ("auto_reset", C.c_uint32), ("max_episode_steps", C.c_uint32), ("rng_kind", C.c_uint32), ("reserved0", C.c_uint32),
        ("seed", C.c_uint64), ("env_id_base", C.c_uint64),
    ]


assert C.sizeof(BatchCfg) == 136


class ObsBuffers(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "scalars", "leaked_credentials", "credential_cache_matrix", "discovered_nodes_properties",
        "nodes_privilegelevel", "mask_local", "mask_remote", "mask_connect", "mask_discrete")]


class WrapperBuffers(C.Structure):
    """mcbs_wrapper_buffers (include/mcbs.h): device arrays of AttackerEnvWrapper's per-env bookkeeping"""
    _fields_ = [(n, C.c_void_p) for n in ("invalid", "reward", "terminated", "timesteps", "valid_action_count", "invalid_action_count",
                                          "episode_returns", "last_cyber_reward", "has_cyber_reward", "rewards", "truncated", "dones",
                                          "episode_return_out", "episode_length_out", "n_done", "executed")]


class DefenderWrapperBuffers(C.Structure):
    """mcbs_defender_wrapper_buffers (include/mcbs.h)"""
    _fields_ = [(n, C.c_void_p) for n in ("valid", "availability", "evicted", "attacker_has_cyber_reward", "attacker_last_cyber_reward",
                                          "timesteps", "valid_action_count", "invalid_action_count", "has_breached_sla", "prev_availability",
                                          "reward", "terminated", "truncated", "breached", "won")]


class DefenderWrapperCfg(C.Structure):
    _fields_ = [("invalid_action_penalty", C.c_double), ("loss_reward", C.c_double), ("sla_worsening_penalty_scale", C.c_double),
                ("maintain_sla", C.c_double), ("winning_reward", C.c_double), ("reset_on_constraint_broken", C.c_int32),
                ("max_timesteps", C.c_int32)]


class DefenderObs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("infected_nodes", "incoming_firewall_status", "outgoing_firewall_status", "services_status")]


class InfoBuffers(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("network_availability", "step_count", "truncated", "out_of_bound", "raw_reward")]


STATE_HEADER_DT = np.dtype([
    ("step_count", "<u4"), ("done", "<u4"), ("truncated", "<u4"), ("episode", "<u4"),
    ("n_discovered", "<u4"), ("n_creds", "<u4"), ("last_outcome_kind", "<u4"), ("last_escalation", "<u4"),
    ("last_new_nodes", "<u4"), ("last_new_creds", "<u4"), ("last_oob", "<u4"), ("pad0", "<u4"),
    ("cum_reward", "<f8"), ("availability", "<f8")])
STATE_NODE_DT = np.dtype([
    ("discovered_props", "<u8"), ("attacked_ever", "<u4"), ("attacked_since", "<u4"),
    ("discovered", "u1"), ("installed", "u1"), ("ever_owned", "u1"), ("running", "u1"),
    ("privilege", "u1"), ("tags", "u1"), ("countdown", "u1"), ("pad", "u1"), ("pad1", "<u4", (2,))])
assert STATE_HEADER_DT.itemsize == 64 and STATE_NODE_DT.itemsize == 32


def state_record_bytes(n_nodes: int, max_creds: int) -> int:
    n = 64 + 32 * n_nodes + 2 * n_nodes + 2 * max_creds
    return (n + 15) & ~15


def split_state(buf: np.ndarray, n_envs: int, n_nodes: int, max_creds: int):
    """View a get_state buffer as (headers[E], nodes[E,N], discovery_order[E,N], credential_cache[E,C])."""
    rb = state_record_bytes(n_nodes, max_creds)
    raw = np.frombuffer(buf, dtype=np.uint8).reshape(n_envs, rb)
    hdr = raw[:, :64].copy().view(STATE_HEADER_DT).reshape(n_envs)
    nodes = raw[:, 64:64 + 32 * n_nodes].copy().view(STATE_NODE_DT).reshape(n_envs, n_nodes)
    o = 64 + 32 * n_nodes
    order = raw[:, o:o + 2 * n_nodes].copy().view("<u2").reshape(n_envs, n_nodes)
    cache = raw[:, o + 2 * n_nodes:o + 2 * n_nodes + 2 * max_creds].copy().view("<u2").reshape(n_envs, max_creds)
    return hdr, nodes, order, cache


@dataclass
class EnvSpec:
    """CyberBattleEnv constructor arguments (cyberbattle_env.py:470-485) plus batching knobs;
    lowered to mcbs_batch_cfg."""
    n_envs: int = 1
    maximum_total_credentials: int = 1000
    maximum_node_count: int = 100
    maximum_discoverable_credentials_per_action: int = 5
    # AttackerGoal (cyberbattle_env.py:227-241); None = no attacker goal
    attacker_goal: Optional[dict] = field(default_factory=lambda: dict(reward=0.0, low_availability=1.0, own_atleast=0, own_atleast_percent=1.0))
    defender_goal_eviction: bool = True
    maintain_sla: float = 0.0
    winning_reward: float = 5000.0
    losing_reward: float = 0.0
    # in-env defender: None or ("scan_and_reimage", probability, scan_capacity, scan_frequency)
    defender: Optional[tuple] = None
    auto_reset: bool = False
    max_episode_steps: int = 0
    rng_kind: int = RNG_PHILOX
    seed: int = 0
    env_id_base: int = 0
    device: int = 0

    def to_cfg(self) -> BatchCfg:
        c = BatchCfg()
        c.abi_version = ABI_VERSION
        c.n_envs = self.n_envs
        c.device = self.device
        c.maximum_node_count = self.maximum_node_count
        c.maximum_total_credentials = self.maximum_total_credentials
        c.maximum_discoverable_credentials_per_action = self.maximum_discoverable_credentials_per_action
        g = self.attacker_goal
        c.has_attacker_goal = 0 if g is None else 1
        if g is not None:
            c.goal_reward = float(g.get("reward", 0.0))
            c.goal_low_availability = float(g.get("low_availability", 1.0))
            c.goal_own_atleast = int(g.get("own_atleast", 0))
            c.goal_own_atleast_percent = float(g.get("own_atleast_percent", 1.0))
        c.defender_goal_eviction = int(bool(self.defender_goal_eviction))
        c.maintain_sla = float(self.maintain_sla)
        c.winning_reward = float(self.winning_reward)
        c.losing_reward = float(self.losing_reward)
        if self.defender is None:
            c.defender_kind = DEFENDER_NONE
            c.scan_frequency = 1
        else:
            if self.defender[0] == "external":       # learned defender acting through mcbs_defender_step
                c.defender_kind = DEFENDER_EXTERNAL
                c.scan_frequency = 1
                return self._finish(c)
            if self.defender[0] == "random_events":  # ExternalRandomEvents (defender.py:58-148): no parameters
                c.defender_kind = DEFENDER_RANDOM_EVENTS
                c.scan_frequency = 1
                return self._finish(c)
            kind, p, cap, freq = self.defender
            if kind != "scan_and_reimage":
                raise ValueError(f"unsupported in-env defender {kind!r}")
            if int(freq) <= 0 or int(cap) < 0:
                raise ValueError("scan_frequency must be positive and scan_capacity non-negative")
            c.defender_kind = DEFENDER_SCAN_AND_REIMAGE
            c.scan_probability, c.scan_capacity, c.scan_frequency = float(p), int(cap), int(freq)
        return self._finish(c)

    def _finish(self, c: BatchCfg) -> BatchCfg:
        c.auto_reset = int(bool(self.auto_reset))
        c.max_episode_steps = int(self.max_episode_steps)
        c.rng_kind = int(self.rng_kind)
        c.seed = int(self.seed) & (2 ** 64 - 1)
        c.env_id_base = int(self.env_id_base)
        return c


class RowCopies(C.Structure):
    """mcbs_row_copies (include/mcbs.h): up to eight (src, dst, row_bytes) triples for mcbs_copy_rows_masked."""
    _fields_ = [("n", C.c_uint32), ("pad", C.c_uint32), ("src", C.c_void_p * 8), ("dst", C.c_void_p * 8), ("row_bytes", C.c_size_t * 8)]
