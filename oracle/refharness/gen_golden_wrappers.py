"""Golden traces of marlon's attacker wrappers, captured from the UNMODIFIED reference
(marlon/baseline_models/env_wrappers/attack_wrapper.py, action_masking.py) over the reference CyberBattleEnv.
Build container only:  python oracle/refharness/gen_golden_wrappers.py   ->  tests/golden/wrap_*.npz
Data only: per step the wrapper-level action (MultiDiscrete(10) row or Discrete index), reward, terminated,
truncated, invalid flag, every flat observation field, CRC32 of the Discrete action mask, defender draw tape.
"""
from __future__ import annotations

import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (loads the reference, installs the defender tape readers)

ref = G.ref
from marlon.baseline_models.env_wrappers.attack_wrapper import AttackerEnvWrapper  # noqa: E402
from marlon.baseline_models.env_wrappers.action_masking import MaskedDiscreteAttackerWrapper  # noqa: E402

FLAT = ["leaked_credentials", "credential_cache_matrix", "discovered_nodes_properties", "nodes_privilegelevel",
        "local_vulnerability", "remote_vulnerability", "connect"]
SCALARS = ["newly_discovered_nodes_count", "lateral_move", "customer_data_found", "probe_result", "escalation",
           "credential_cache_length", "discovered_node_count"]


def snap(obs):
    out = {k: np.array(obs[k]) for k in FLAT}
    out["scalars"] = np.array([int(obs[k]) for k in SCALARS], np.int32)
    return out


def run(name, make_cyber_env, spec, steps, seed, discrete, max_timesteps, tape_dps):
    rng = np.random.Generator(np.random.PCG64(seed))
    cyber = make_cyber_env()
    w = AttackerEnvWrapper(cyber, max_timesteps=max_timesteps, invalid_action_reward_modifier=-1)
    m = MaskedDiscreteAttackerWrapper(w)
    nvec = np.asarray(w.action_space.nvec)
    obs, _ = m.reset(seed=seed)
    rec = {k: [] for k in ["action", "reward", "terminated", "truncated", "invalid", "mask_crc", "mask_sum", "tape", "was_reset"] + FLAT + ["scalars"]}
    rec_reset = {k: [] for k in FLAT + ["scalars"]}
    first = snap(obs)
    for t in range(steps):
        mask = m.action_masks()
        if discrete:
            if rng.random() < 0.8:
                a = int(rng.choice(np.flatnonzero(mask)))           # what MaskablePPO would sample from
            else:
                a = int(rng.integers(0, m.action_space.n))
        else:
            a = (rng.random(10) * nvec).astype(np.int64)
            nd = int(obs["discovered_node_count"])
            if rng.random() < 0.75:                                  # mostly in range, so that the env moves
                for i in (1, 3, 4, 6, 7):
                    a[i] = rng.integers(0, nd)
                a[9] = rng.integers(0, max(1, int(obs["credential_cache_length"])))
        tape = list(rng.random(tape_dps)) if tape_dps else []
        if tape_dps:
            G.TAPE.load(tape + [0.0] * 8)
        obs, reward, terminated, truncated, info = (m.step(np.int64(a)) if discrete else w.step(a))
        s = snap(obs)
        rec["action"].append(a)
        rec["reward"].append(float(reward))
        rec["terminated"].append(int(terminated))
        rec["truncated"].append(int(truncated))
        rec["invalid"].append(int(bool(info.get("invalid_action", False))))
        am = m.action_masks()
        rec["mask_crc"].append(zlib.crc32(am.astype(np.int8).tobytes()))
        rec["mask_sum"].append(int(am.sum()))
        rec["tape"].append(tape)
        for k in FLAT + ["scalars"]:
            rec[k].append(s[k])
        done = terminated or truncated
        rec["was_reset"].append(int(done))
        if done:
            obs, _ = m.reset(seed=seed + 1000 * (t + 1))
            r = snap(obs)
            for k in FLAT + ["scalars"]:
                rec_reset[k].append(r[k])
    out = {k: np.asarray(v) for k, v in rec.items()}
    out["tape"] = np.asarray(rec["tape"], np.float64).reshape(steps, -1)
    out["mask_crc"] = out["mask_crc"].astype(np.uint32)
    for k in FLAT + ["scalars"]:
        out["first_" + k] = first[k]
        if rec_reset[k]:
            out["after_reset_" + k] = np.asarray(rec_reset[k])
    spec = dict(spec, discrete=bool(discrete), max_timesteps=max_timesteps)
    out["spec_json"] = np.frombuffer(json.dumps(spec).encode(), dtype=np.uint8)
    path = os.path.join(G.GOLDEN, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name:28s} steps={steps} reward_sum={out['reward'].sum():8.1f} invalid={out['invalid'].sum():4d} "
          f"dones={out['was_reset'].sum():3d} size={os.path.getsize(path) / 1024:.0f} KiB")


def main():
    AG, DC, SAR = ref.env.AttackerGoal, ref.env.DefenderConstraint, ref.defender.ScanAndReimageCompromisedMachines

    def goal(**kw):
        g = dict(reward=0.0, low_availability=1.0, own_atleast=0, own_atleast_percent=1.0)
        g.update(kw)
        return g
    sp_t = dict(maximum_node_count=12, maximum_total_credentials=10, maximum_discoverable_credentials_per_action=5,
                attacker_goal=goal(own_atleast=6), winning_reward=5000.0, losing_reward=0.0, maintain_sla=0.80,
                defender=["scan_and_reimage", 0.6, 2, 5])

    def toyctf_def():
        return ref.CyberBattleToyCtf(attacker_goal=AG(own_atleast=6), defender_agent=SAR(0.6, 2, 5),
                                     defender_constraint=DC(maintain_sla=0.80), maximum_node_count=12,
                                     maximum_total_credentials=10, throws_on_invalid_actions=False)
    run("wrap_toyctf_md_s61", toyctf_def, sp_t, 260, 61, False, 120, 4)
    run("wrap_toyctf_discrete_s62", toyctf_def, sp_t, 260, 62, True, 120, 4)
    sp_c = dict(maximum_node_count=12, maximum_total_credentials=12, maximum_discoverable_credentials_per_action=5,
                attacker_goal=goal(), winning_reward=5000.0, losing_reward=0.0, maintain_sla=0.0, defender=None)

    def chain10():
        return ref.CyberBattleChain(size=10, attacker_goal=AG(own_atleast_percent=1.0), maximum_node_count=12,
                                    maximum_total_credentials=12, throws_on_invalid_actions=False)
    run("wrap_chain10_md_s63", chain10, sp_c, 300, 63, False, 100, 0)
    run("wrap_chain10_discrete_s64", chain10, sp_c, 300, 64, True, 100, 0)


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


# ---------------------------------------------------------------------------------------------------------------------
# Learned defender: DefenderEnvWrapper + LearningDefender next to the attacker wrapper on ONE CyberBattleEnv, as
# MultiAgentUniverse.build wires them (multiagent_universe.py:160-199).  Quirk Q14: both defender objects capture
# `environment` / `_defender_actuator` at construction while CyberBattleEnv.reset() replaces them, so after the first
# reset the reference's defender acts on a dead copy.  The traces are captured with the defender RE-BOUND to the live
# objects after every reset (instance attributes set from outside, source untouched): the behaviour the code intends and
# the one this build implements.
def run_defender(name, make_cyber_env, spec, steps, seed, max_timesteps):
    from marlon.baseline_models.env_wrappers.defend_wrapper import DefenderEnvWrapper
    from marlon.baseline_models.env_wrappers.environment_event_source import EnvironmentEventSource
    rng = np.random.Generator(np.random.PCG64(seed))
    cyber = make_cyber_env()
    events = EnvironmentEventSource()
    aw = AttackerEnvWrapper(cyber, event_source=events, max_timesteps=max_timesteps, invalid_action_reward_modifier=-1)
    dw = DefenderEnvWrapper(cyber, attacker_reward_store=aw, event_source=events, defender=True, max_timesteps=max_timesteps,
                            invalid_action_reward=-1, reset_on_constraint_broken=True, loss_reward=-5000.0)

    def rebind():
        dw._actuator = cyber._defender_actuator
        dw.defender._actuator = cyber._defender_actuator
        dw.defender._environment = cyber.environment

    def reset_both(s):
        obs, _ = aw.reset(seed=s)
        rebind()
        dobs, _ = dw.reset()          # reset_request was raised by the attacker's reset: no second notification; resets the env once
                                      # more (as in the reference) and the defender's counters
        rebind()
        dw._prev_network_availability = float(dw._actuator.network_availability)
        aw._last_transformed_observation = aw.transform_observation(cyber.reset(seed=s)[0])
        rebind()
        return dw.observe()

    nvec_a, nvec_d = np.asarray(aw.action_space.nvec), np.asarray(dw.action_space.nvec)
    dobs = reset_both(seed)
    keys = ["infected_nodes", "incoming_firewall_status", "outgoing_firewall_status", "services_status"]
    rec = {k: [] for k in ["a_action", "a_reward", "a_terminated", "a_truncated", "a_invalid", "d_action", "d_reward", "d_terminated",
                           "d_truncated", "d_valid", "d_availability", "was_reset", "a_discovered"] + ["d_" + k for k in keys]}
    first = {k: np.asarray(dobs[k], np.int8) for k in keys}
    nodes = list(cyber.environment.network.nodes)
    for t in range(steps):
        a = (rng.random(10) * nvec_a).astype(np.int64)
        nd = len(cyber._CyberBattleEnv__discovered_nodes)
        if rng.random() < 0.8:
            for i in (1, 3, 4, 6, 7):
                a[i] = rng.integers(0, nd)
            a[9] = rng.integers(0, max(1, len(cyber._CyberBattleEnv__credential_cache)))
        obs, ar, aterm, atrunc, ainfo = aw.step(a)
        rec["a_action"].append(a); rec["a_reward"].append(float(ar)); rec["a_terminated"].append(int(aterm))
        rec["a_truncated"].append(int(atrunc)); rec["a_invalid"].append(int(bool(ainfo.get("invalid_action", False))))
        rec["a_discovered"].append(int(obs["discovered_node_count"]))
        d = (rng.random(12) * nvec_d).astype(np.int64)
        r = rng.random()
        if r < 0.35:                                  # bias toward re-imaging owned nodes so that the attacker feels it
            owned = [i for i, n in enumerate(nodes) if cyber.environment.get_node(n).agent_installed]
            d[0] = 0
            d[1] = int(rng.choice(owned)) if owned and rng.random() < 0.7 else d[1]
        done = bool(aterm or atrunc)
        if not done:
            dobs, dr, dterm, dtrunc, _ = dw.step(d)
            rec["d_action"].append(d); rec["d_reward"].append(float(dr)); rec["d_terminated"].append(int(dterm))
            rec["d_truncated"].append(int(dtrunc)); rec["d_valid"].append(int(bool(dw.last_action_valid)))
            rec["d_availability"].append(float(dw.last_availability))
            for k in keys:
                rec["d_" + k].append(np.asarray(dobs[k], np.int8))
            done = bool(dterm or dtrunc)
        else:
            rec["d_action"].append(np.full(12, -2, np.int64)); rec["d_reward"].append(0.0); rec["d_terminated"].append(0)
            rec["d_truncated"].append(0); rec["d_valid"].append(0); rec["d_availability"].append(float(cyber._defender_actuator.network_availability))
            for k in keys:
                rec["d_" + k].append(np.asarray(dw.observe()[k], np.int8))
        rec["was_reset"].append(int(done))
        if done:
            reset_both(seed + 1000 * (t + 1))
    out = {k: np.asarray(v) for k, v in rec.items()}
    for k in keys:
        out["first_" + k] = first[k]
    out["spec_json"] = np.frombuffer(json.dumps(dict(spec, max_timesteps=max_timesteps)).encode(), dtype=np.uint8)
    path = os.path.join(G.GOLDEN, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name:28s} steps={steps} a_reward={out['a_reward'].sum():8.1f} d_reward={out['d_reward'].sum():9.1f} "
          f"d_valid={out['d_valid'].sum():4d} resets={out['was_reset'].sum():3d} size={os.path.getsize(path) / 1024:.0f} KiB")


def main_defender():
    AG, DC = ref.env.AttackerGoal, ref.env.DefenderConstraint
    sp = dict(maximum_node_count=12, maximum_total_credentials=10, maximum_discoverable_credentials_per_action=5,
              attacker_goal=dict(reward=0.0, low_availability=1.0, own_atleast=6, own_atleast_percent=1.0),
              winning_reward=5000.0, losing_reward=-5000.0, maintain_sla=0.60, defender=["external"])

    def toyctf_marl():   # multiagent_universe.py:160-165: no in-env defender_agent, SLA 0.60, losing_reward -5000
        return ref.CyberBattleToyCtf(attacker_goal=AG(own_atleast=6), defender_constraint=DC(maintain_sla=0.60), losing_reward=-5000.0,
                                     maximum_node_count=12, maximum_total_credentials=10, throws_on_invalid_actions=False)
    run_defender("wrap_defender_toyctf_s71", toyctf_marl, sp, 400, 71, 60)
    run_defender("wrap_defender_toyctf_s72", toyctf_marl, sp, 400, 72, 150)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "defender":
        main_defender()


# ---------------------------------------------------------------------------------------------------------------------
# Two-agent EPISODES in the call order of marl_algorithm.run_episode (marlon/baseline_models/multiagent/marl_algorithm.py:176-250;
# that module itself cannot be imported: it needs Stable-Baselines3).  Each agent's `.env` there is a DummyVecEnv, which resets a
# wrapper inside step_wait as soon as it reports done; that auto-reset is what makes the defender's step AFTER an attacker `done`
# return `-last attacker reward` with truncated=True (reset_request protocol: attack_wrapper.py:433-435, defend_wrapper.py:269-271,
# 479-482).  The loop below issues exactly those calls on the reference's own wrapper objects (defender re-bound to the live
# environment after every reset, as above) and records what they return.
def run_two_agent_episodes(name, make_cyber_env, spec, n_episodes, seed, max_timesteps, max_steps, p_in_range=0.85):
    from marlon.baseline_models.env_wrappers.defend_wrapper import DefenderEnvWrapper
    from marlon.baseline_models.env_wrappers.environment_event_source import EnvironmentEventSource
    rng = np.random.Generator(np.random.PCG64(seed))
    cyber = make_cyber_env()
    events = EnvironmentEventSource()
    aw = AttackerEnvWrapper(cyber, event_source=events, max_timesteps=max_timesteps, invalid_action_reward_modifier=-1)
    dw = DefenderEnvWrapper(cyber, attacker_reward_store=aw, event_source=events, defender=True, max_timesteps=max_timesteps,
                            invalid_action_reward=-1, reset_on_constraint_broken=True, loss_reward=-5000.0)

    def rebind():
        dw._actuator = cyber._defender_actuator
        dw.defender._actuator = cyber._defender_actuator
        dw.defender._environment = cyber.environment

    def attacker_env_reset():      # DummyVecEnv.reset / the auto-reset inside step_wait
        aw.reset()
        rebind()

    def defender_env_reset():
        dw.reset()
        rebind()
        dw._prev_network_availability = float(dw._actuator.network_availability)
        dw.last_availability = dw._prev_network_availability

    nvec_a, nvec_d = np.asarray(aw.action_space.nvec), np.asarray(dw.action_space.nvec)
    nodes = list(cyber.environment.network.nodes)
    rec = {k: [] for k in ["episode", "a_action", "a_reward", "a_done", "d_action", "d_reward", "d_done", "d_terminated", "d_truncated"]}
    ends = []
    for ep in range(n_episodes):
        attacker_env_reset()                     # run_episode: attacker_agent.env.reset()
        dw.on_reset(0)                           #              defender_agent.wrapper.on_reset(0)
        defender_env_reset()                     #              defender_agent.env.reset()
        aw._last_transformed_observation = aw.transform_observation(cyber.reset()[0])   # (the attacker's cached observation of the twice-reset env)
        rebind()
        n_steps, reason = 0, "max_steps"
        while n_steps < max_steps:
            a = (rng.random(10) * nvec_a).astype(np.int64)
            nd = len(cyber._CyberBattleEnv__discovered_nodes)
            if rng.random() < p_in_range:
                for i in (1, 3, 4, 6, 7):
                    a[i] = rng.integers(0, nd)
                a[9] = rng.integers(0, max(1, len(cyber._CyberBattleEnv__credential_cache)))
            _, r1, term1, trunc1, _ = aw.step(a)
            dones1 = bool(term1 or trunc1)
            if dones1:
                attacker_env_reset()             # DummyVecEnv auto-reset: notifies the defender (reset_request, last reward)
            d = (rng.random(12) * nvec_d).astype(np.int64)
            if rng.random() < 0.3:
                owned = [i for i, n in enumerate(nodes) if cyber.environment.get_node(n).agent_installed]
                d[0] = 0
                d[1] = int(rng.choice(owned)) if owned and rng.random() < 0.7 else d[1]
            _, r2, term2, trunc2, _ = dw.step(d)
            dones2 = bool(term2 or trunc2)
            if dones2:
                defender_env_reset()
            rec["episode"].append(ep); rec["a_action"].append(a); rec["a_reward"].append(float(r1)); rec["a_done"].append(int(dones1))
            rec["d_action"].append(d); rec["d_reward"].append(float(r2)); rec["d_done"].append(int(dones2))
            rec["d_terminated"].append(int(bool(term2))); rec["d_truncated"].append(int(bool(trunc2)))
            if dones1 or dones2:
                reason = ("attacker" if dones1 else "") + ("+" if dones1 and dones2 else "") + ("defender" if dones2 else "")
                break
            n_steps += 1
        ends.append(reason)
    out = {k: np.asarray(v) for k, v in rec.items()}
    out["spec_json"] = np.frombuffer(json.dumps(dict(spec, max_timesteps=max_timesteps, max_steps=max_steps, ends=ends)).encode(), dtype=np.uint8)
    path = os.path.join(G.GOLDEN, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name:28s} episodes={n_episodes} steps={len(out['a_reward'])} ends={ {e: ends.count(e) for e in set(ends)} } "
          f"a_reward={out['a_reward'].sum():.1f} d_reward={out['d_reward'].sum():.1f} size={os.path.getsize(path) / 1024:.0f} KiB")


def main_episodes():
    AG, DC = ref.env.AttackerGoal, ref.env.DefenderConstraint
    sp = dict(maximum_node_count=12, maximum_total_credentials=10, maximum_discoverable_credentials_per_action=5,
              attacker_goal=dict(reward=0.0, low_availability=1.0, own_atleast=6, own_atleast_percent=1.0),
              winning_reward=5000.0, losing_reward=-5000.0, maintain_sla=0.60, defender=["external"])

    def toyctf_marl():
        return ref.CyberBattleToyCtf(attacker_goal=AG(own_atleast=6), defender_constraint=DC(maintain_sla=0.60), losing_reward=-5000.0,
                                     maximum_node_count=12, maximum_total_credentials=10, throws_on_invalid_actions=False)
    run_two_agent_episodes("wrap_episode_toyctf_s73", toyctf_marl, sp, 14, 73, 40, 200)
    run_two_agent_episodes("wrap_episode_toyctf_s74", toyctf_marl, sp, 10, 74, 300, 25)
    # short episodes, half of the attacker's actions out of range: the attacker's LAST reward is often -1, which the defender's final
    # step must return negated (reset_request)
    run_two_agent_episodes("wrap_episode_toyctf_s75", toyctf_marl, sp, 30, 75, 12, 100, p_in_range=0.5)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "episodes":
    main_episodes()
