"""Gym-shaped front ends of the step engine: the drop-in surface of `CyberBattleEnv`.

Mirrors, over the C ABI, the interface the reference exposes at
src/CyberBattleSim/cyberbattle/_env/cyberbattle_env.py — constructor keywords :470-485, `reset` :1187,
`step` :1145, `compute_action_mask` :679, `is_node_owned` :1009, `is_action_valid` :1016,
`sample_valid_action` :1041, `bounds` :404, `identifiers` :400 — and the thin topology subclasses
`CyberBattleChain` (_env/cyberbattle_chain.py:10-19) and `CyberBattleToyCtf` (_env/cyberbattle_toyctf.py:8-12).

`CyberBattleEnv` here is the single-environment facade (a batch of one on the GPU): it exists so that code and
tests written against the reference read the same; throughput comes from `marlon_amd.wrappers.AttackerVecEnv` /
`marlon_amd.engine.BatchEngine`, which advance tens of thousands of environments per launch.  No rule of the
simulation is implemented in this file — every transition, observation field and mask is computed by the HIP
kernels; this file converts between the reference's Python types and device buffers.
"""
from __future__ import annotations

from typing import Dict, List, NamedTuple, Optional, Tuple

import numpy as np

from . import flatten as _flatten
from . import model
from ._abi import RNG_PHILOX, RNG_TAPE, EnvSpec
from .samples import chainpattern, toy_ctf


class AttackerGoal(NamedTuple):
    """cyberbattle_env.py:227-241"""
    reward: float = 0.0
    low_availability: float = 1.0
    own_atleast: int = 0
    own_atleast_percent: float = 1.0


class DefenderGoal(NamedTuple):
    """cyberbattle_env.py:244-248"""
    eviction: bool


class DefenderConstraint(NamedTuple):
    """cyberbattle_env.py:251-254"""
    maintain_sla: float


class EnvironmentBounds(NamedTuple):
    """cyberbattle_env.py:172-224"""
    maximum_total_credentials: int
    maximum_node_count: int
    maximum_discoverable_credentials_per_action: int
    port_count: int
    property_count: int
    local_attacks_count: int
    remote_attacks_count: int


class DefenderAgent:
    """Base of in-env defenders (_env/defender.py:18-24).  Only declarative defenders the kernels implement can be
    attached: they carry parameters, not a Python `step`."""


class ScanAndReimageCompromisedMachines(DefenderAgent):
    """_env/defender.py:27-55: every `scan_frequency` steps scan `scan_capacity` random nodes, detect an installed
    agent with `probability`, re-image the node if it is re-imagable."""

    def __init__(self, probability: float, scan_capacity: int, scan_frequency: int):
        self.probability = probability
        self.scan_capacity = scan_capacity
        self.scan_frequency = scan_frequency


class ExternalRandomEvents(DefenderAgent):
    """_env/defender.py:58-148: every step, on every node, with probability 0.1 each: patch a vulnerability away, stop a service,
    plant a library vulnerability, remove a firewall rule, add an ALLOW rule on a random common port.  No parameters; the draws
    come from the batch's Philox stream (or a tape, for parity with the reference's global generators)."""


class OutOfBoundIndexError(Exception):
    """cyberbattle_env.py:135-136 (swallowed inside step, kept for API symmetry)"""


def spec_from_kwargs(n_envs: int, maximum_total_credentials: int, maximum_node_count: int,
                     maximum_discoverable_credentials_per_action: int, defender_agent, attacker_goal, defender_goal,
                     defender_constraint, winning_reward: float, losing_reward: float, **extra) -> EnvSpec:
    if defender_agent is not None and not isinstance(defender_agent, (ScanAndReimageCompromisedMachines, ExternalRandomEvents)):
        raise NotImplementedError(
            f"in-env defender {type(defender_agent).__name__} is not implemented on the device "
            "(supported: ScanAndReimageCompromisedMachines, ExternalRandomEvents)")
    goal = None if attacker_goal is None else dict(attacker_goal._asdict())
    if defender_agent is None:
        d = None
    elif isinstance(defender_agent, ExternalRandomEvents):
        d = ("random_events",)
    else:
        d = ("scan_and_reimage", defender_agent.probability, defender_agent.scan_capacity, defender_agent.scan_frequency)
    return EnvSpec(n_envs=n_envs, maximum_total_credentials=maximum_total_credentials, maximum_node_count=maximum_node_count,
                   maximum_discoverable_credentials_per_action=maximum_discoverable_credentials_per_action,
                   attacker_goal=goal, defender_goal_eviction=bool(defender_goal.eviction),
                   maintain_sla=float(defender_constraint.maintain_sla), winning_reward=float(winning_reward),
                   losing_reward=float(losing_reward), defender=d, **extra)


OBS_FIELDS = ["scalars", "leaked_credentials", "credential_cache_matrix", "discovered_nodes_properties",
              "nodes_privilegelevel", "mask_local", "mask_remote", "mask_connect"]
SCALAR_KEYS = ["newly_discovered_nodes_count", "lateral_move", "customer_data_found", "probe_result", "escalation",
               "credential_cache_length", "discovered_node_count"]


class _ActionSpaceSeeds:
    """Holder of the sampler generator the reference keeps on its DiscriminatedUnion action space
    (_env/discriminatedunion.py:42-47): `env.action_space.union_np_random`."""

    def __init__(self, seed=None):
        self.union_np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))

    def seed(self, seed=None):
        self.union_np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
        return [seed]


class CyberBattleEnv:
    """One environment with the reference's reset()/step() surface, stepped on the GPU."""

    metadata = {"render_modes": ["human"]}
    privilege_levels = model.PrivilegeLevel.MAXIMUM + 1

    def __init__(self, initial_environment: model.Environment, maximum_total_credentials: int = 1000,
                 maximum_node_count: int = 100, maximum_discoverable_credentials_per_action: int = 5,
                 defender_agent: Optional[DefenderAgent] = None,
                 attacker_goal: Optional[AttackerGoal] = AttackerGoal(own_atleast_percent=1.0),
                 defender_goal=DefenderGoal(eviction=True), defender_constraint=DefenderConstraint(maintain_sla=0.0),
                 winning_reward=5000.0, losing_reward=0.0, renderer="", observation_padding=True,
                 throws_on_invalid_actions=True, device: Optional[str] = None, seed: int = 0, draw_tape: bool = False):
        from .engine import BatchEngine
        if not observation_padding:
            raise NotImplementedError("observation_padding=False is not supported (gym>=0.26 requires padding, cyberbattle_env.py:498-500)")
        self.__initial_environment = initial_environment
        self.topo = _flatten.flatten(initial_environment)
        self.spec = spec_from_kwargs(1, maximum_total_credentials, maximum_node_count, maximum_discoverable_credentials_per_action,
                                     defender_agent, attacker_goal, defender_goal, defender_constraint, winning_reward, losing_reward,
                                     seed=seed, rng_kind=RNG_TAPE if draw_tape else RNG_PHILOX)
        self.__throws = throws_on_invalid_actions
        self.__WINNING_REWARD, self.__LOSING_REWARD = winning_reward, losing_reward
        self.__defender_agent = defender_agent
        self.__defender_constraint, self.__defender_goal, self.__attacker_goal = defender_constraint, defender_goal, attacker_goal
        self._engine = BatchEngine(self.topo, self.spec, device=device)      # raises ValueError like validate_environment
        ids = initial_environment.identifiers
        self.__bounds = EnvironmentBounds(maximum_total_credentials, maximum_node_count, maximum_discoverable_credentials_per_action,
                                          len(ids.ports), len(ids.properties), len(ids.local_vulnerabilities),
                                          len(ids.remote_vulnerabilities))
        self.__node_count = self.topo.n_nodes
        self.__done = False
        self.__episode_rewards: List[float] = []
        self.np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(None)))
        self.action_space = _ActionSpaceSeeds()
        self._obs = self._engine.alloc_obs(OBS_FIELDS)
        self.__owned_cache: Optional[List[int]] = None
        self.__refresh_lists()

    # ---- properties the reference exposes ----
    @property
    def name(self) -> str:
        return "CyberBattleEnv"

    @property
    def identifiers(self) -> model.Identifiers:
        return self.__initial_environment.identifiers

    @property
    def bounds(self) -> EnvironmentBounds:
        return self.__bounds

    @property
    def unwrapped(self):
        return self

    @property
    def environment(self) -> model.Environment:
        """`CyberBattleEnv.environment` (cyberbattle_env.py:371), as a SNAPSHOT: a copy of the initial model whose nodes carry the dynamic
        fields the device state holds — agent_installed, privilege_level, status (Running / Imaging) and the privilege_k tags appended to
        node.properties (actions.py:378).  Read-only: the state lives on the GPU, edits of the copy do not reach it (marlon's defender
        acts through DefenderVecEnv / mcbs_defender_step instead of mutating this object)."""
        import copy
        env = copy.deepcopy(self.__initial_environment)
        _, nodes, _, _ = self._engine.get_state()
        for i, (node_id, info) in enumerate(env.nodes()):
            info.agent_installed = bool(nodes["installed"][0][i])
            info.privilege_level = model.PrivilegeLevel(int(nodes["privilege"][0][i]))
            info.status = model.MachineStatus.Running if nodes["running"][0][i] else model.MachineStatus.Imaging
            for k in range(4):
                tag = f"privilege_{k}"
                if (int(nodes["tags"][0][i]) >> k) & 1 and tag not in info.properties:
                    info.properties.append(tag)
        return env

    def __defender_goal_reached(self) -> bool:
        """`__defender_goal_reached` (cyberbattle_env.py:1112-1116): eviction asked for and no node left with privilege >= LocalUser."""
        return bool(self.__defender_goal.eviction) and not bool((self.__privilege_by_node >= 1).any())

    # ---- helpers ----
    def __refresh_lists(self) -> None:
        hdr, nodes, order, cache = self._engine.get_state()
        n, c = int(hdr["n_discovered"][0]), int(hdr["n_creds"][0])
        self.__discovered_nodes = [self.topo.node_ids[i] for i in order[0, :n]]
        self.__credential_cache = [model.CachedCredential(*self.topo.triples[i]) for i in cache[0, :c]]
        self.__privilege_by_node = nodes["privilege"][0].astype(int)
        self.__installed_by_node = nodes["installed"][0].astype(bool)

    def __to_observation(self) -> Dict[str, object]:
        o = {k: v[0].cpu().numpy() for k, v in self._obs.items()}
        obs: Dict[str, object] = {}
        for i, k in enumerate(SCALAR_KEYS):
            obs[k] = np.int32(o["scalars"][i]) if i < 5 else int(o["scalars"][i])
        obs["leaked_credentials"] = tuple(o["leaked_credentials"])
        obs["credential_cache_matrix"] = tuple(o["credential_cache_matrix"])
        obs["discovered_nodes_properties"] = o["discovered_nodes_properties"]
        obs["nodes_privilegelevel"] = o["nodes_privilegelevel"]
        obs["action_mask"] = {"local_vulnerability": o["mask_local"], "remote_vulnerability": o["mask_remote"], "connect": o["mask_connect"]}
        obs["_discovered_nodes"] = self.__discovered_nodes
        obs["_explored_network"] = None      # networkx rendering data, not produced
        return obs

    def __info(self) -> Dict[str, object]:
        i = self._engine.step_info()
        return dict(description="CyberBattle simulation", duration_in_ms=0.0, step_count=int(i["step_count"][0]),
                    network_availability=float(i["network_availability"][0]), credential_cache=self.__credential_cache)

    @staticmethod
    def __row(action: Dict[str, np.ndarray]) -> List[int]:
        assert len(action) == 1
        if "local_vulnerability" in action:
            a = action["local_vulnerability"]
            return [0, int(a[0]), int(a[1]), 0, 0]
        if "remote_vulnerability" in action:
            a = action["remote_vulnerability"]
            return [1, int(a[0]), int(a[1]), int(a[2]), 0]
        if "connect" in action:
            a = action["connect"]
            return [2, int(a[0]), int(a[1]), int(a[2]), int(a[3])]
        raise ValueError("Invalid discriminated union value: " + str(action))

    # ---- gym surface ----
    def reset(self, *, seed: Optional[int] = None, options: Optional[dict] = None):
        self._engine.reset()
        self.np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
        self.__done = False
        self.__episode_rewards = []
        self.__owned_cache = None
        self._engine.observe(self._obs)
        self.__refresh_lists()
        obs = self.__to_observation()
        obs["credential_cache_length"] = 0
        info = self.__info()
        info["duration_in_ms"] = 0
        return obs, info

    def set_draw_tape(self, draws) -> None:
        """Defender draws for the next step (parity with the reference's global RNGs)."""
        self._engine.set_draw_tape(np.asarray(draws, np.float64).reshape(1, -1))

    def step(self, action: Dict[str, np.ndarray]):
        if self.__done:
            raise RuntimeError("new episode must be started with env.reset()")
        row = self.__row(action)
        if self.__throws:
            self.__raise_if_invalid(row)
        reward, term = self._engine.step_observe(np.asarray([row], np.int32), self._obs)
        reward, done = float(reward[0]), bool(term[0])
        oob = bool(self._engine.info["out_of_bound"][0])
        self.__done = done
        self.__episode_rewards.append(reward)
        self.__refresh_lists()
        if not oob:
            self.__owned_cache = None          # the reference keeps its stale cache on the out-of-bound path (quirk Q8)
        obs = self.__to_observation()
        return obs, reward, done, False, self.__info()

    def __raise_if_invalid(self, row: List[int]) -> None:
        """throws_on_invalid_actions=True (actions.py:444-448,486-490,547-563): the checks that raise instead of -1."""
        kind, a, b = row[0], row[1], row[2]
        nd = len(self.__discovered_nodes)
        if kind == 2 and not (0 <= row[4] < len(self.__credential_cache)):
            return
        if not (0 <= a < nd) or (kind != 0 and not (0 <= b < nd)):
            return                              # out-of-bound path, handled inside step
        src = self.__discovered_nodes[a]
        if not self.__installed_by_node[self.topo.node_ids.index(src)]:
            raise ValueError(f"Agent does not owned the source node '{src}'" if kind else f"Agent does not owned the node '{src}'")

    def compute_action_mask(self) -> Dict[str, np.ndarray]:
        m = self._engine.action_mask(self._engine.alloc_obs(["mask_local", "mask_remote", "mask_connect"]))
        return {"local_vulnerability": m["mask_local"][0].cpu().numpy(), "remote_vulnerability": m["mask_remote"][0].cpu().numpy(),
                "connect": m["mask_connect"][0].cpu().numpy()}

    def apply_mask(self, action, mask=None) -> bool:
        if mask is None:
            mask = self.compute_action_mask()
        kind = next(iter(action.keys()))
        return bool(mask[kind][tuple(int(x) for x in action[kind])])

    def is_node_owned(self, node: int) -> bool:
        if node < 0 or node >= len(self.__discovered_nodes):
            raise OutOfBoundIndexError(f"Node index ({node}) is invalid; only {len(self.__discovered_nodes)} nodes discovered so far.")
        return bool(self.__privilege_by_node[self.topo.node_ids.index(self.__discovered_nodes[node])] > 0)

    def is_action_valid(self, action, action_mask=None) -> bool:
        kind = next(iter(action.keys()))
        a = [int(x) for x in action[kind]]
        nd, b = len(self.__discovered_nodes), self.__bounds
        if kind == "local_vulnerability":
            ok = a[0] < nd and self.is_node_owned(a[0]) and a[1] < b.local_attacks_count
        elif kind == "remote_vulnerability":
            ok = a[0] < nd and self.is_node_owned(a[0]) and a[1] < nd and a[2] < b.remote_attacks_count
        else:
            ok = a[0] < nd and self.is_node_owned(a[0]) and a[1] < nd and a[2] < b.port_count and a[3] < len(self.__credential_cache)
        return bool(ok and self.apply_mask(action, action_mask))

    # ---- samplers: same draws from the same generators as cyberbattle_env.py:935-1055 ----
    def __owned_indices(self) -> List[int]:
        if self.__owned_cache is None:       # network order of the owned nodes, as external indices (:832-838)
            self.__owned_cache = [self.__discovered_nodes.index(n) for i, n in enumerate(self.topo.node_ids)
                                  if self.__privilege_by_node[i] >= 1]
        return self.__owned_cache

    def sample_connect_action_in_expected_range(self):
        if len(self.__credential_cache) <= 0:
            raise ValueError("Cannot sample a connect action until the agent discovers more potential target nodes.")
        r = self.np_random
        return {"connect": np.array([r.choice(self.__owned_indices()), r.integers(0, len(self.__discovered_nodes)),
                                     r.integers(0, self.__bounds.port_count), r.integers(0, len(self.__credential_cache))], np.int32)}

    def sample_action_in_range(self, kinds: Optional[List[int]] = None):
        if kinds is None:
            kinds = [0, 1, 2]
        if len(self.__credential_cache) == 0:
            kinds = [t for t in kinds if t != 2]
        assert kinds, "Kinds list cannot be empty"
        r = self.action_space.union_np_random
        kind = r.choice(kinds)
        if kind == 2:
            return self.sample_connect_action_in_expected_range()
        if kind == 1:                          # sic: 1 draws a LOCAL action in the reference (:985-994)
            return {"local_vulnerability": np.array([r.choice(self.__owned_indices()), r.integers(0, self.__bounds.local_attacks_count)], np.int32)}
        return {"remote_vulnerability": np.array([r.choice(self.__owned_indices()), r.integers(0, len(self.__discovered_nodes)),
                                                  r.integers(0, self.__bounds.remote_attacks_count)], np.int32)}

    def sample_valid_action(self, kinds=None):
        mask = self.compute_action_mask()
        action = self.sample_action_in_range(kinds)
        while not self.apply_mask(action, mask):
            action = self.sample_action_in_range(kinds)
        return action

    def close(self) -> None:
        self._engine.close()


class CyberBattleChain(CyberBattleEnv):
    """_env/cyberbattle_chain.py:10-19"""

    def __init__(self, size, **kwargs):
        self.size = size
        super().__init__(initial_environment=chainpattern.new_environment(size), **kwargs)

    @property
    def name(self) -> str:
        return f"CyberBattleChain-{self.size}"


class CyberBattleToyCtf(CyberBattleEnv):
    """_env/cyberbattle_toyctf.py:8-12"""

    def __init__(self, **kwargs):
        super().__init__(initial_environment=toy_ctf.new_environment(), **kwargs)


class CyberBattleTiny(CyberBattleEnv):
    """_env/cyberbattle_tiny.py:8-12 (`CyberBattleTiny-v0`)"""

    def __init__(self, **kwargs):
        from .samples import tinytoy
        super().__init__(initial_environment=tinytoy.new_environment(), **kwargs)


class CyberBattleRandom(CyberBattleEnv):
    """_env/cyberbattle_random.py:10-14 (`CyberBattleRandom-v0`): a freshly generated 50-client / 3 x 15-server traffic
    network with up to 32 credentials per leak.  `seed` (not in the reference, whose networks are unrepeatable) makes the
    generated network reproducible; further keyword arguments go to CyberBattleEnv."""

    def __init__(self, seed=None, **kwargs):
        from .samples import generate_network
        kwargs.setdefault("maximum_discoverable_credentials_per_action", 32)
        super().__init__(initial_environment=generate_network.new_environment(n_servers_per_protocol=15, seed=seed), **kwargs)


class CyberBattleActiveDirectory(CyberBattleEnv):
    """_env/active_directory.py:6-10 (`ActiveDirectory-v{seed}`)"""

    def __init__(self, seed, **kwargs):
        from .samples import active_directory
        super().__init__(initial_environment=active_directory.new_random_environment(seed), **kwargs)


class CyberBattleActiveDirectoryTiny(CyberBattleEnv):
    """_env/active_directory.py:13-15 (`ActiveDirectoryTiny-v0`)"""

    def __init__(self, **kwargs):
        from .samples import active_directory
        super().__init__(initial_environment=active_directory.new_tiny_environment(), **kwargs)
