"""GPU tests of the host-side mirrors of the reference interface:
the gym-shaped single-env facade and the batched counterparts of marlon's attacker wrappers, against golden traces
captured from the reference's own classes."""
import json
import os
import zlib

import numpy as np
import pytest

from tests import parity

pytestmark = pytest.mark.gpu

GOLDEN = parity.GOLDEN


def _env_for(name, sj, **kw):
    from marlon_amd import cyberbattle_env as ce
    from marlon_amd import model
    from marlon_amd.samples import kitchen_sink
    g = sj["attacker_goal"]
    d = sj["defender"]
    common = dict(maximum_node_count=sj["maximum_node_count"], maximum_total_credentials=sj["maximum_total_credentials"],
                  maximum_discoverable_credentials_per_action=sj["maximum_discoverable_credentials_per_action"],
                  attacker_goal=ce.AttackerGoal(**g), defender_constraint=ce.DefenderConstraint(maintain_sla=sj["maintain_sla"]),
                  defender_agent=None if d is None else (ce.ExternalRandomEvents() if d[0] == "random_events" else
                                                         ce.ScanAndReimageCompromisedMachines(d[1], d[2], d[3])),
                  winning_reward=sj["winning_reward"], losing_reward=sj["losing_reward"], throws_on_invalid_actions=False,
                  draw_tape=d is not None)
    common.update(kw)
    if name.startswith("chain100"):
        return ce.CyberBattleChain(size=100, **common)
    if name.startswith("chain10"):
        return ce.CyberBattleChain(size=10, **common)
    if name.startswith("toyctf"):
        return ce.CyberBattleToyCtf(**common)
    if name.startswith("tiny_"):
        return ce.CyberBattleTiny(**common)
    if name[:2] == "ad" and name[2].isdigit():
        return ce.CyberBattleActiveDirectory(seed=int(name[2]), **common)
    if name.startswith("random_s"):
        return ce.CyberBattleRandom(seed=int(name.split("_")[1][1:]), **common)
    if name.startswith("sink_evict"):
        return ce.CyberBattleEnv(kitchen_sink.build(model, entry_reimagable=True), **common)
    return ce.CyberBattleEnv(kitchen_sink.build(model), **common)


VALID_TRACES = ["chain10_valid_s1", "chain10_valid_s2", "chain10_rewardgoal_s6", "toyctf_defender_s11", "toyctf_defender_s12",
                "toyctf_slabreak_s16", "sink_evict_s44", "sink_attackerwin_s45", "chain100_defender_s31",
                # the other registered environments, through their own facade classes (CyberBattleTiny / ActiveDirectory / Random)
                "tiny_defender_s62", "ad0_valid_s64", "ad1_valid_s69", "random_s4_valid_s66"]


@pytest.mark.parametrize("name", VALID_TRACES)
def test_gym_facade_reproduces_reference_episode_from_seeds(name):
    """Same seeds as the harness -> sample_valid_action draws the reference's actions, step returns its results:
    the whole loop `env.reset(seed); a = env.sample_valid_action(); env.step(a)` is interchangeable."""
    z, sj = parity.load_trace(name)
    seed = int(name.rsplit("_s", 1)[1])
    env = _env_for(name, sj)
    env.action_space.union_np_random = np.random.Generator(np.random.PCG64(seed + 1))
    obs, info = env.reset(seed=seed)
    np.testing.assert_array_equal(obs["discovered_nodes_properties"], z["reset_discovered_nodes_properties"])
    np.testing.assert_array_equal(obs["action_mask"]["local_vulnerability"], z["reset_mask_local"])
    assert info["network_availability"] == 1.0 and obs["credential_cache_length"] == 0
    episode = 0
    T = min(len(z["reward"]), 150 if name.startswith("chain100") else 400)
    for t in range(T):
        a = env.sample_valid_action()
        row = [0, *a["local_vulnerability"], 0, 0] if "local_vulnerability" in a else \
              [1, *a["remote_vulnerability"], 0] if "remote_vulnerability" in a else [2, *a["connect"]]
        assert [int(x) for x in row] == z["actions"][t].tolist(), f"{name} step {t}: sampled action differs from the reference's"
        if z["tape"].size:
            env.set_draw_tape(z["tape"][t])
        obs, reward, done, truncated, info = env.step(a)
        assert reward == z["reward"][t] and done == bool(z["terminated"][t]) and truncated is False
        assert info["step_count"] == z["step_count"][t]
        assert np.float64(info["network_availability"]).view(np.uint64) == z["availability"][t].view(np.uint64)
        assert [int(obs[k]) for k in ("newly_discovered_nodes_count", "lateral_move", "customer_data_found", "probe_result", "escalation",
                                       "credential_cache_length", "discovered_node_count")] == z["scalars"][t].tolist()
        np.testing.assert_array_equal(np.stack(obs["leaked_credentials"]), z["leaked_credentials"][t])
        np.testing.assert_array_equal(obs["nodes_privilegelevel"], z["nodes_privilegelevel"][t])
        assert [env.topo.node_ids.index(n) for n in obs["_discovered_nodes"]] == z["order"][t][:z["n_order"][t]].tolist()
        if done:
            with pytest.raises(RuntimeError, match=r"new episode must be started with env\.reset\(\)"):
                env.step(a)
            episode += 1
            env.reset(seed=seed + 1000 * episode)
    env.close()


def test_gym_facade_accepts_external_random_events():
    """CyberBattleToyCtf(defender_agent=ExternalRandomEvents()) runs on the device; availability drops as services stop."""
    from marlon_amd import cyberbattle_env as ce
    env = ce.CyberBattleToyCtf(attacker_goal=ce.AttackerGoal(own_atleast_percent=1.0), defender_agent=ce.ExternalRandomEvents(),
                               defender_constraint=ce.DefenderConstraint(maintain_sla=0.0), maximum_node_count=12,
                               maximum_total_credentials=10, throws_on_invalid_actions=False, seed=5)
    env.reset(seed=1)
    lowest = 1.0
    for _ in range(150):
        obs, reward, done, truncated, info = env.step(env.sample_valid_action())
        lowest = min(lowest, info["network_availability"])
        if done:
            env.reset()
    assert lowest < 1.0
    env.close()


def test_gym_facade_errors_and_helpers():
    from marlon_amd import cyberbattle_env as ce
    with pytest.raises(ValueError, match=r"Network node count \(12\) exceeds the specified limit of 10"):
        ce.CyberBattleChain(size=10, maximum_node_count=10, maximum_total_credentials=12)
    env = ce.CyberBattleChain(size=10, maximum_node_count=12, maximum_total_credentials=12)      # throws_on_invalid_actions=True
    obs, _ = env.reset(seed=0)
    assert env.bounds.port_count == 8 and env.bounds.local_attacks_count == 5 and env.name == "CyberBattleChain-10"
    assert env.is_node_owned(0) and env.compute_action_mask()["local_vulnerability"][0].tolist() == [0, 1, 0, 0, 0]
    obs, r, done, _, _ = env.step({"local_vulnerability": np.array([0, 1])})
    assert r == 14.0 and obs["discovered_node_count"] == 2 and not env.is_node_owned(1)
    with pytest.raises(ValueError, match="Agent does not owned the node '1_LinuxNode'"):
        env.step({"local_vulnerability": np.array([1, 0])})
    with pytest.raises(ValueError, match="Agent does not owned the source node '1_LinuxNode'"):
        env.step({"remote_vulnerability": np.array([1, 0, 0])})
    assert env.is_action_valid({"connect": np.array([0, 1, 2, 0])}) and not env.is_action_valid({"connect": np.array([0, 1, 2, 1])})
    obs, r, done, _, _ = env.step({"local_vulnerability": np.array([5, 0])})                     # out-of-bound: blank observation
    assert r == 0.0 and (obs["discovered_nodes_properties"] == 2).all() and obs["action_mask"]["connect"].sum() == 0
    assert env.compute_action_mask()["connect"].sum() > 0                                        # compute_action_mask is never blank
    env.close()


WRAP = ["wrap_toyctf_md_s61", "wrap_toyctf_discrete_s62", "wrap_chain10_md_s63", "wrap_chain10_discrete_s64",
        # round 3 (oracle/refharness/gen_golden_bounds.py): observation bounds other than the tight ones — Chain-4 @ 9/7, ToyCtf @ 11/7, Chain-10 @ 14/16
        "wrap_chain4_discrete_b9x7_s97", "wrap_toyctf_md_b11x7_s98", "wrap_chain10_discrete_b14x16_s99"]
FLAT = {"leaked_credentials": "leaked_credentials", "credential_cache_matrix": "credential_cache_matrix",
        "discovered_nodes_properties": "discovered_nodes_properties", "nodes_privilegelevel": "nodes_privilegelevel",
        "local_vulnerability": "local_vulnerability", "remote_vulnerability": "remote_vulnerability", "connect": "connect"}


def _check_flat(obs, z, prefix, t, ctx):
    from marlon_amd.cyberbattle_env import SCALAR_KEYS
    ref = (lambda k: z[prefix + k]) if t is None else (lambda k: z[prefix + k][t])
    assert [int(obs[k][0]) for k in SCALAR_KEYS] == ref("scalars").tolist(), ctx + " scalars"
    for k in FLAT:
        np.testing.assert_array_equal(obs[k][0].cpu().numpy().reshape(-1), np.asarray(ref(k)).reshape(-1), err_msg=f"{ctx} {k}")


@pytest.mark.parametrize("name", WRAP)
def test_attacker_vec_env_matches_marlon_wrappers(name):
    """AttackerVecEnv (n_envs=1) == AttackerEnvWrapper / MaskedDiscreteAttackerWrapper of the reference, step by step:
    decode, interception of undiscovered node indices, reward modifier, truncation, auto-reset, flat observation, mask."""
    from marlon_amd import cyberbattle_env as ce
    from marlon_amd._abi import RNG_TAPE
    from marlon_amd.wrappers import AttackerVecEnv
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    sj = json.loads(bytes(z["spec_json"]).decode())
    topo = parity.topology_for(name[len("wrap_"):])
    d = sj["defender"]
    env = AttackerVecEnv(topo, 1, maximum_node_count=sj["maximum_node_count"], maximum_total_credentials=sj["maximum_total_credentials"],
                         attacker_goal=ce.AttackerGoal(**sj["attacker_goal"]), defender_constraint=ce.DefenderConstraint(sj["maintain_sla"]),
                         defender_agent=None if d is None else (ce.ExternalRandomEvents() if d[0] == "random_events" else
                                                         ce.ScanAndReimageCompromisedMachines(d[1], d[2], d[3])),
                         max_timesteps=sj["max_timesteps"], discrete=sj["discrete"], rng_kind=RNG_TAPE)
    _check_flat(env.observation, z, "first_", None, name + " reset")
    resets = 0
    for t in range(len(z["reward"])):
        if z["tape"].size:
            env.engine.set_draw_tape(z["tape"][t:t + 1])
        a = np.asarray(z["action"][t]).reshape((1,) if sj["discrete"] else (1, 10))
        obs, r, term, trunc, info = env.step(a)
        ctx = f"{name} step {t}"
        assert float(r[0]) == z["reward"][t], ctx + f" reward {float(r[0])} != {z['reward'][t]}"
        assert int(term[0]) == z["terminated"][t] and int(trunc[0]) == z["truncated"][t], ctx + " flags"
        assert int(info["invalid_action"][0]) == z["invalid"][t], ctx + " invalid"
        if z["was_reset"][t]:
            _check_flat(env.terminal_observation, z, "", t, ctx + " terminal")
            _check_flat(obs, z, "after_reset_", resets, ctx + " after reset")
            resets += 1
        else:
            _check_flat(obs, z, "", t, ctx)
            m = env.action_masks()[0].cpu().numpy()
            assert int(m.sum()) == z["mask_sum"][t] and zlib.crc32(m.astype(np.int8).tobytes()) == z["mask_crc"][t], ctx + " action mask"
    env.close()


def test_attacker_vec_env_batch_against_oracle():
    """4 096 envs, MultiDiscrete actions drawn on the host, wrapper semantics reproduced with the oracle as checker."""
    from marlon_amd import cyberbattle_env as ce
    from marlon_amd.wrappers import AttackerVecEnv
    from oracle.oracle import Oracle
    topo = parity.topology_for("toyctf")
    E, T = 4096, 80
    env = AttackerVecEnv(topo, E, maximum_node_count=12, maximum_total_credentials=10, attacker_goal=ce.AttackerGoal(own_atleast=6),
                         defender_agent=ce.ScanAndReimageCompromisedMachines(0.6, 2, 5), defender_constraint=ce.DefenderConstraint(0.8),
                         max_timesteps=50, seed=11)
    orc = Oracle(topo, env.spec)
    rng = np.random.Generator(np.random.PCG64(3))
    timesteps = np.zeros(E, np.int64)
    n_disc = np.ones(E, np.int64)
    for t in range(T):
        a = (rng.random((E, 10)) * env.nvec).astype(np.int64)
        fix = rng.random(E) < 0.7
        for i in (1, 3, 4, 6, 7):
            a[fix, i] = (rng.random(fix.sum()) * n_disc[fix]).astype(np.int64)
        kind = a[:, 0]
        src = np.where(kind == 0, a[:, 1], np.where(kind == 1, a[:, 3], a[:, 6]))
        tgt = np.where(kind == 0, 0, np.where(kind == 1, a[:, 4], a[:, 7]))
        valid = (src < n_disc) & ((kind == 0) | (tgt < n_disc))
        rows = np.zeros((E, 5), np.int32)
        rows[:, 0] = np.where(valid, kind, 3)
        rows[:, 1] = src
        rows[:, 2] = np.where(kind == 0, a[:, 2], tgt)
        rows[:, 3] = np.where(kind == 1, a[:, 5], np.where(kind == 2, a[:, 8], 0))
        rows[:, 4] = np.where(kind == 2, a[:, 9], 0)
        obs, r, term, trunc, info = env.step(a)
        o = orc.step(rows)
        timesteps += 1
        exp_r = o["reward"] + np.where(valid, 0.0, -1.0)
        np.testing.assert_array_equal(r.double().cpu().numpy(), exp_r, err_msg=f"step {t} reward")
        np.testing.assert_array_equal(term.cpu().numpy(), o["terminated"], err_msg=f"step {t} terminated")
        np.testing.assert_array_equal(trunc.cpu().numpy(), (timesteps >= 50).astype(np.uint8), err_msg=f"step {t} truncated")
        np.testing.assert_array_equal(info["invalid_action"].cpu().numpy(), ~valid)
        dones = (o["terminated"] != 0) | (timesteps >= 50)
        for i in np.flatnonzero(dones):
            orc.reset(int(i))
        timesteps[dones] = 0
        n_disc = obs["discovered_node_count"].cpu().numpy().astype(np.int64)
        _, _, order, _ = orc.get_state()
        np.testing.assert_array_equal(n_disc, (order != 0xFFFF).sum(axis=1), err_msg=f"step {t} discovered count after reset")
    env.close()


# ---------------------------------------------------------------- learned defender
DEF_KEYS = ["infected_nodes", "incoming_firewall_status", "outgoing_firewall_status", "services_status"]


@pytest.mark.parametrize("name", ["wrap_defender_toyctf_s71", "wrap_defender_toyctf_s72"])
def test_defender_vec_env_matches_marlon_defender_wrapper(name):
    """AttackerVecEnv + DefenderVecEnv on one batch == AttackerEnvWrapper + DefenderEnvWrapper/LearningDefender on one
    CyberBattleEnv (defender re-bound to the live environment, quirk Q14): validity, firewall edits, re-imaging and its
    effect on the attacker, availability, the shaped reward (fp64), SLA termination, eviction, truncation, observation."""
    from marlon_amd import cyberbattle_env as ce
    from marlon_amd.wrappers import AttackerVecEnv, DefenderVecEnv
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    sj = json.loads(bytes(z["spec_json"]).decode())
    att = AttackerVecEnv(parity.topology_for("toyctf"), 1, maximum_node_count=12, maximum_total_credentials=10,
                         attacker_goal=ce.AttackerGoal(**sj["attacker_goal"]), defender_constraint=ce.DefenderConstraint(sj["maintain_sla"]),
                         losing_reward=sj["losing_reward"], max_timesteps=sj["max_timesteps"], auto_reset=False, learned_defender=True)
    dfd = DefenderVecEnv(att, max_timesteps=sj["max_timesteps"], invalid_action_reward=-1, loss_reward=-5000.0)
    for k in DEF_KEYS:
        np.testing.assert_array_equal(dfd.observation[k][0].cpu().numpy(), z["first_" + k], err_msg=f"{name} first {k}")
    for t in range(len(z["a_reward"])):
        ctx = f"{name} step {t}"
        obs, r, term, trunc, info = att.step(z["a_action"][t].reshape(1, 10))
        assert float(r[0]) == z["a_reward"][t], f"{ctx}: attacker reward {float(r[0])} != {z['a_reward'][t]}"
        assert int(term[0]) == z["a_terminated"][t] and int(trunc[0]) == z["a_truncated"][t], ctx + " attacker flags"
        assert int(info["invalid_action"][0]) == z["a_invalid"][t] and int(obs["discovered_node_count"][0]) == z["a_discovered"][t], ctx
        if z["d_action"][t][0] > -2:
            dobs, dr, dterm, dtrunc, dinfo = dfd.step(z["d_action"][t].reshape(1, 12))
            assert int(dinfo["valid_action"][0]) == z["d_valid"][t], ctx + f" validity of {z['d_action'][t].tolist()}"
            assert float(dinfo["network_availability"][0]) == z["d_availability"][t], ctx + " availability"
            assert float(dr[0]) == z["d_reward"][t], f"{ctx}: defender reward {float(dr[0])!r} != {z['d_reward'][t]!r}"
            assert int(dterm[0]) == z["d_terminated"][t] and int(dtrunc[0]) == z["d_truncated"][t], ctx + " defender flags"
            for k in DEF_KEYS:
                np.testing.assert_array_equal(dobs[k][0].cpu().numpy(), z["d_" + k][t], err_msg=f"{ctx} {k}")
        if z["was_reset"][t]:
            att.reset()
            dfd.reset()
    att.close()


@pytest.mark.parametrize("case", ["toyctf", "toyctf_separate_observation", "random24", "random70"])
def test_defender_step_batch_against_oracle(case, monkeypatch):
    """Random attacker rows and random defender vectors on a batch that does not fill its last workgroup: validity, availability bits,
    eviction, the four observation fields and the attacker's rewards equal the oracle's (which keeps real rule lists) at every step.
    ToyCtf and a 24-node instance of the config-5 generator write the observation from the turn kernel itself (round 3: one launch per
    turn), MCBS_NO_FUSED_DEFENDER_OBS=1 and the 70-node instance (two words per node set) through the separate observation launch;
    buffers pre-filled with a sentinel.  defend_wrapper.py:329-412,492-534, defender.py:31-107."""
    from marlon_amd import flatten as F, model
    from marlon_amd._abi import EnvSpec
    from marlon_amd.samples import random_net
    from oracle.oracle import Oracle
    if case.startswith("toyctf"):
        topo, nm, cm, E, T = parity.topology_for("toyctf"), 12, 10, 2048 + 37, 120
    else:
        n = int(case[6:])
        topo = F.flatten(random_net.build(model, n, 5))
        nm, cm, E, T = n, max(1, len(topo.triples)), 300 + 37, 60
    spec = EnvSpec(n_envs=E, maximum_node_count=nm, maximum_total_credentials=cm, maximum_discoverable_credentials_per_action=8,
                   attacker_goal=dict(own_atleast=6, own_atleast_percent=1.0),
                   maintain_sla=0.6, losing_reward=-5000.0, defender=("external",), auto_reset=True, max_episode_steps=80, seed=4)
    if case == "toyctf_separate_observation":
        monkeypatch.setenv("MCBS_NO_FUSED_DEFENDER_OBS", "1")
    eng = _engine_mod().BatchEngine(topo, spec)
    monkeypatch.delenv("MCBS_NO_FUSED_DEFENDER_OBS", raising=False)
    orc = Oracle(topo, spec)
    rng = np.random.Generator(np.random.PCG64(9))
    N = int(topo.n_nodes)
    nvec = np.array([5, N, N, 6, 2, N, 6, 2, N, 3, N, 3])
    dobs = eng.alloc_defender_obs()
    for v in dobs.values():
        v.fill_(9)
    for t in range(T):
        a = eng.sample_actions(t % 4 != 0, seed=2, step=t)
        r, d = eng.step(a)
        o = orc.step(a.cpu().numpy())
        np.testing.assert_array_equal(r.double().cpu().numpy(), o["reward"], err_msg=f"step {t} attacker reward")
        np.testing.assert_array_equal(d.cpu().numpy(), o["terminated"], err_msg=f"step {t} terminated")
        da = (rng.random((E, 12)) * nvec).astype(np.int64)
        da[rng.random(E) < 0.05, 0] = -1
        da[rng.random(E) < 0.05, 0] = -2
        v, av, ev = eng.defender_step(da, dobs)
        od = orc.defender_step(da)
        np.testing.assert_array_equal(v.cpu().numpy(), od["valid"], err_msg=f"step {t} valid")
        np.testing.assert_array_equal(av.cpu().numpy().view(np.uint64), od["availability"].view(np.uint64), err_msg=f"step {t} availability")
        np.testing.assert_array_equal(ev.cpu().numpy(), od["evicted"], err_msg=f"step {t} evicted")
        oo = orc.defender_observe()
        for k in DEF_KEYS:
            np.testing.assert_array_equal(dobs[k].cpu().numpy(), oo[k], err_msg=f"step {t} {k}")
    eng.close()


def _engine_mod():
    from marlon_amd import engine
    return engine


def test_gym_facade_exposes_what_marlons_callers_read():
    """The attributes marlon's wrappers and simulation helpers read through name mangling (attack_wrapper.py:71-72,118;
    defend_wrapper.py:52,260-261; multiagent/simulation.py:22,36): discovered nodes, winning / losing reward, defender constraint,
    defender goal test, episode rewards, and `environment` as a snapshot of the live model."""
    from marlon_amd import cyberbattle_env as ce
    from marlon_amd import model
    env = ce.CyberBattleChain(size=4, maximum_node_count=6, maximum_total_credentials=6, throws_on_invalid_actions=False, losing_reward=-7.0,
                              defender_constraint=ce.DefenderConstraint(maintain_sla=0.6))
    env.reset(seed=0)
    assert env._CyberBattleEnv__WINNING_REWARD == 5000.0 and env._CyberBattleEnv__LOSING_REWARD == -7.0
    assert env._CyberBattleEnv__defender_constraint.maintain_sla == 0.6
    assert env._CyberBattleEnv__discovered_nodes == ["start"] and env._CyberBattleEnv__episode_rewards == []
    assert env._CyberBattleEnv__defender_goal_reached() is False             # the start node is owned
    snap = env.environment
    assert isinstance(snap, model.Environment) and [n for n, _ in snap.nodes()][0] == "start"
    start = snap.get_node("start")
    assert start.agent_installed and start.privilege_level >= model.PrivilegeLevel.LocalUser and start.status == model.MachineStatus.Running
    assert not snap.get_node("1_LinuxNode").agent_installed
    obs, r, done, _, _ = env.step({"local_vulnerability": np.array([0, 1])})   # ScanExplorerRecentFiles: discovers 1_LinuxNode
    assert env._CyberBattleEnv__discovered_nodes == ["start", "1_LinuxNode"] and env._CyberBattleEnv__episode_rewards == [r]
    snap.get_node("start").agent_installed = False                           # a snapshot: editing it does not touch the device state
    assert env.environment.get_node("start").agent_installed
    env.close()


@pytest.mark.parametrize("n_nodes,masks", [(12, True), (12, False), (16, False), (24, True), (24, False), (70, True), (70, False)])
def test_attacker_vec_env_on_larger_topologies_against_oracle(n_nodes, masks):
    """The batched attacker wrapper beyond the reference's sample topologies (the config-5 generator at 12 / 16 nodes: general state layout
    with the sixteen-lanes-per-env observation kernels and the three-launch step; at 24 and 70 nodes: general state layout,
    one and two words per set, action spaces too large for the fused mask writers, flat-mask rows padded to whole cache lines), with the
    oracle as checker of the wrapper's semantics: MultiDiscrete actions drawn on the host (a share of them with undiscovered node indices:
    intercepted), in-env ScanAndReimage, truncation and auto-reset — rewards, flags, interception, the small observation fields and (when
    materialised) the Discrete action mask of every env that stepped and did not end.  attack_wrapper.py:255-372, action_masking.py:90-110."""
    from marlon_amd import cyberbattle_env as ce, flatten as F, model
    from marlon_amd.samples import random_net
    from marlon_amd.wrappers import AttackerVecEnv
    from oracle.oracle import Oracle
    topo = F.flatten(random_net.build(model, n_nodes, 31))
    E, T, MAXT = 64, 70, 25
    Cm = max(1, len(topo.triples))
    env = AttackerVecEnv(topo, E, maximum_node_count=n_nodes, maximum_total_credentials=Cm, maximum_discoverable_credentials_per_action=8,
                         attacker_goal=ce.AttackerGoal(own_atleast_percent=0.7), defender_agent=ce.ScanAndReimageCompromisedMachines(0.5, 2, 3),
                         defender_constraint=ce.DefenderConstraint(0.3), max_timesteps=MAXT, seed=13, materialize_masks=masks)
    orc = Oracle(topo, env.spec)
    small = ["scalars", "leaked_credentials", "credential_cache_matrix", "discovered_nodes_properties", "nodes_privilegelevel"]
    mask_fields = ["mask_connect", "mask_local", "mask_remote"] if masks else []
    from marlon_amd.cyberbattle_env import SCALAR_KEYS
    rng = np.random.Generator(np.random.PCG64(9))
    timesteps = np.zeros(E, np.int64)
    n_disc = np.ones(E, np.int64)
    checked = ended = 0
    for t in range(T):
        a = (rng.random((E, 10)) * env.nvec).astype(np.int64)
        fix = rng.random(E) < 0.8
        for i in (1, 3, 4, 6, 7):
            a[fix, i] = (rng.random(fix.sum()) * n_disc[fix]).astype(np.int64)
        kind = a[:, 0]
        src = np.where(kind == 0, a[:, 1], np.where(kind == 1, a[:, 3], a[:, 6]))
        tgt = np.where(kind == 0, 0, np.where(kind == 1, a[:, 4], a[:, 7]))
        valid = (src < n_disc) & ((kind == 0) | (tgt < n_disc))
        rows = np.zeros((E, 5), np.int32)
        rows[:, 0] = np.where(valid, kind, 3)
        rows[:, 1] = src
        rows[:, 2] = np.where(kind == 0, a[:, 2], tgt)
        rows[:, 3] = np.where(kind == 1, a[:, 5], np.where(kind == 2, a[:, 8], 0))
        rows[:, 4] = np.where(kind == 2, a[:, 9], 0)
        obs, r, term, trunc, info = env.step(a)
        oo = orc.alloc_obs(small + mask_fields)
        o = orc.step(rows, obs=oo)
        timesteps += 1
        ctx = f"random_net({n_nodes}) masks={masks} step {t}"
        np.testing.assert_array_equal(r.double().cpu().numpy(), o["reward"] + np.where(valid, 0.0, -1.0), err_msg=ctx + " reward")
        np.testing.assert_array_equal(term.cpu().numpy(), o["terminated"], err_msg=ctx + " terminated")
        np.testing.assert_array_equal(trunc.cpu().numpy(), (timesteps >= MAXT).astype(np.uint8), err_msg=ctx + " truncated")
        np.testing.assert_array_equal(info["invalid_action"].cpu().numpy(), ~valid, err_msg=ctx + " interception")
        dones = (o["terminated"] != 0) | (timesteps >= MAXT)
        keep = valid & ~dones                            # these envs' observation is this step's (pre-defender) observation
        sel = np.flatnonzero(keep)
        if sel.size:
            got_scalars = np.stack([obs[k].cpu().numpy() for k in SCALAR_KEYS], axis=1)
            np.testing.assert_array_equal(got_scalars[sel], oo["scalars"][sel], err_msg=ctx + " scalars")
            for k in small[1:]:
                np.testing.assert_array_equal(obs[k].cpu().numpy().reshape(E, -1)[sel], oo[k].reshape(E, -1)[sel], err_msg=f"{ctx} {k}")
            if masks:
                flat = np.concatenate([oo["mask_connect"].reshape(E, -1), oo["mask_local"].reshape(E, -1), oo["mask_remote"].reshape(E, -1)], axis=1)
                np.testing.assert_array_equal(env.action_masks().cpu().numpy()[sel], flat[sel] != 0, err_msg=ctx + " action mask")
            checked += sel.size
        for i in np.flatnonzero(dones):
            orc.reset(int(i))
        ended += int(dones.sum())
        timesteps[dones] = 0
        n_disc = obs["discovered_node_count"].cpu().numpy().astype(np.int64)
    assert checked > E * T // 3 and ended >= E
    env.close()
