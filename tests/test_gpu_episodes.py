"""Episode drivers (marlon_amd/simulate.py) = marlon.simulate -> marl_algorithm.run_episode (marl_algorithm.py:144-252).

* `run_episode` against two-agent episodes captured from the reference's own AttackerEnvWrapper / DefenderEnvWrapper /
  LearningDefender in run_episode's call order (tests/golden/wrap_episode_*.npz, oracle/refharness/gen_golden_wrappers.py
  `episodes`): both reward traces step by step, which side ended each episode, the defender's `-last attacker reward` after an
  attacker `done`, the max_steps stop.
* `run_episodes` with the masked-random attacker on Chain-4: the sampled actions replayed through the CPU oracle, rewards and dones
  equal at every step — which also settles how often such an attacker wins within 400 steps."""
import json
import os

import numpy as np
import pytest

from tests import parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["wrap_episode_toyctf_s73", "wrap_episode_toyctf_s74", "wrap_episode_toyctf_s75"])
def test_run_episode_replays_reference_two_agent_episodes(name):
    from marlon_amd import cyberbattle_env as ce
    from marlon_amd.simulate import run_episode
    from marlon_amd.wrappers import AttackerVecEnv, DefenderVecEnv
    z = np.load(os.path.join(parity.GOLDEN, name + ".npz"))
    sj = json.loads(bytes(z["spec_json"]).decode())
    att = AttackerVecEnv(parity.topology_for("toyctf"), 1, maximum_node_count=12, maximum_total_credentials=10,
                         attacker_goal=ce.AttackerGoal(**sj["attacker_goal"]), defender_constraint=ce.DefenderConstraint(sj["maintain_sla"]),
                         losing_reward=sj["losing_reward"], max_timesteps=sj["max_timesteps"], auto_reset=False, learned_defender=True)
    dfd = DefenderVecEnv(att, max_timesteps=sj["max_timesteps"], invalid_action_reward=-1, loss_reward=-5000.0)
    ep = z["episode"]
    ends = {"attacker": 0, "defender": 0, "max_steps": 0}
    for e in range(int(ep.max()) + 1):
        idx = np.flatnonzero(ep == e)
        cur = {"a": 0, "d": 0}

        def pa(env, idx=idx, cur=cur):
            a = z["a_action"][idx[cur["a"]]].reshape(1, 10)
            cur["a"] += 1
            return a

        def pd(env, idx=idx, cur=cur):
            d = z["d_action"][idx[cur["d"]]].reshape(1, 12)
            cur["d"] += 1
            return d
        out = run_episode(att, dfd, pa, pd, max_steps=sj["max_steps"])
        n = len(idx)
        ctx = f"{name} episode {e}"
        assert int(out["lengths"][0]) == n and out["steps"] == n, f"{ctx}: {int(out['lengths'][0])} steps, the reference took {n}"
        np.testing.assert_array_equal(out["attacker_rewards"][:, 0].cpu().numpy(), z["a_reward"][idx], err_msg=ctx + " attacker rewards")
        np.testing.assert_array_equal(out["defender_rewards"][:, 0].cpu().numpy(), z["d_reward"][idx], err_msg=ctx + " defender rewards")
        assert bool(out["attacker_done"][0]) == bool(z["a_done"][idx[-1]]) and bool(out["defender_done"][0]) == bool(z["d_done"][idx[-1]]), ctx
        ends["attacker" if z["a_done"][idx[-1]] else ("defender" if z["d_done"][idx[-1]] else "max_steps")] += 1
        if z["a_done"][idx[-1]]:                       # reset_request rule (defend_wrapper.py:269-271), as the reference returned it
            assert z["d_reward"][idx[-1]] == -z["a_reward"][idx[-1]]
    assert sum(ends.values()) == len(sj["ends"])
    att.close()


def _decode_discrete(idx, N, L, R, P, C):
    """MaskedDiscreteAttackerWrapper._decode (action_masking.py:112-142) on the host -> engine rows (kind, a, b, c, d)."""
    rows = np.zeros((len(idx), 5), np.int32)
    cs, ls = N * N * P * C, N * L
    for i, a in enumerate(idx):
        a = int(a)
        if a < cs:
            q, cred = divmod(a, C)
            q, port = divmod(q, P)
            src, tgt = divmod(q, N)
            rows[i] = (2, src, tgt, port, cred)
        elif a < cs + ls:
            src, v = divmod(a - cs, L)
            rows[i] = (0, src, v, 0, 0)
        else:
            q, v = divmod(a - cs - ls, R)
            src, tgt = divmod(q, N)
            rows[i] = (1, src, tgt, v, 0)
    return rows


def test_run_episodes_random_policy_chain4_replayed_through_the_oracle():
    from marlon_amd import cyberbattle_env as ce
    from marlon_amd.samples import chainpattern
    from marlon_amd.simulate import random_policy, run_episodes
    from marlon_amd.wrappers import AttackerVecEnv
    from oracle.oracle import Oracle
    E, T, MAXT = 512, 400, 400
    env = AttackerVecEnv(chainpattern.new_environment(4), E, maximum_node_count=6, maximum_total_credentials=6,
                         attacker_goal=ce.AttackerGoal(own_atleast_percent=1.0), max_timesteps=MAXT, discrete=True)
    out = run_episodes(env, random_policy(seed=3), max_steps=T, record_actions=True)
    acts, r, d = out["actions"].cpu().numpy(), out["rewards"].cpu().numpy(), out["dones"].cpu().numpy()
    topo = env.topo
    N, Cm = 6, 6
    L, R, P = len(topo.local_vulnerabilities), len(topo.remote_vulnerabilities), len(topo.ports)
    orc = Oracle(topo, env.spec)
    timesteps = np.zeros(E, np.int64)
    wins = 0
    for t in range(T):
        o = orc.step(_decode_discrete(acts[t], N, L, R, P, Cm))
        timesteps += 1
        np.testing.assert_array_equal(r[t].astype(np.float64), o["reward"], err_msg=f"step {t} reward")      # masked actions are never intercepted
        done = (o["terminated"] != 0) | (timesteps >= MAXT)
        np.testing.assert_array_equal(d[t] != 0, done, err_msg=f"step {t} done")
        assert (o["oob"] == 0).all()
        wins += int((o["reward"] == 5000.0).sum())
        for i in np.flatnonzero(done):
            orc.reset(int(i))
        timesteps[done] = 0
    ep = out["episodes"].cpu().numpy()
    assert (ep >= 1).all()                                 # every env ended at least one episode (a win, or truncation at 400)
    assert int((r == 5000.0).sum()) == wins                # the engine and the reference's rules agree on how many attackers won
    assert r.min() >= 0.0 and r.sum() > 0
    print(f"masked-random attacker, Chain-4, {E} envs x {T} steps: {wins} wins, {int(ep.sum())} episodes")
    env.close()


def test_run_episode_batch_with_random_agents():
    """run_episode for 1 024 ToyCtf envs with both random agents: per-env lengths, the ended-by flags and the reward traces are
    consistent (rows past an env's last step are zero; the defender's last reward after an attacker done is the negated attacker
    reward), and the attacker side equals an attacker-only run of the same actions up to the first defender intervention."""
    from marlon_amd import cyberbattle_env as ce
    from marlon_amd.simulate import random_defender_policy, random_policy, run_episode
    from marlon_amd.wrappers import AttackerVecEnv, DefenderVecEnv
    E = 1024
    att = AttackerVecEnv(parity.topology_for("toyctf"), E, maximum_node_count=12, maximum_total_credentials=10,
                         attacker_goal=ce.AttackerGoal(own_atleast=6), defender_constraint=ce.DefenderConstraint(0.6), losing_reward=-5000.0,
                         max_timesteps=50, discrete=True, auto_reset=False, learned_defender=True)
    dfd = DefenderVecEnv(att, max_timesteps=50, invalid_action_reward=-1, loss_reward=-5000.0)
    out = run_episode(att, dfd, random_policy(1), random_defender_policy(2), max_steps=80)
    ar, dr = out["attacker_rewards"].cpu().numpy(), out["defender_rewards"].cpu().numpy()
    n = out["lengths"].cpu().numpy()
    ad, dd = out["attacker_done"].cpu().numpy(), out["defender_done"].cpu().numpy()
    assert ar.shape == dr.shape == (out["steps"], E) and (n >= 1).all() and (n <= 50).all() and n.max() == out["steps"]
    assert (ad | dd).all()                                 # wrapper truncation at 50 < max_steps 80: every episode ended by a done
    for i in range(E):
        assert (ar[n[i]:, i] == 0).all() and (dr[n[i]:, i] == 0).all()
        if ad[i]:
            assert dr[n[i] - 1, i] == -ar[n[i] - 1, i] and dd[i]
    assert (n < 50).any()                                  # some episodes ended early (SLA breach, eviction, attacker win)
    with pytest.raises(ValueError, match="auto_reset=False"):
        run_episode(AttackerVecEnvStub(), None)
    att.close()


class AttackerVecEnvStub:
    auto_reset = True


def test_simulate_entry_point():
    """marlon.simulate.simulate's counterpart: option handling, the default universe (ToyCtf, invalid-action modifiers 0, SLA 0.60 and
    losing reward -5000 with a defender), one episode per env bounded by `timesteps`."""
    from marlon_amd.simulate import simulate
    with pytest.raises(ValueError, match="Attacker cannot be none"):
        simulate(10, "None", "None")
    with pytest.raises(NotImplementedError, match="Stable-Baselines3"):
        simulate(10, "Load", "None", attacker_file="ppo.zip")
    out = simulate(60, "Random", "Random", n_envs=64, seed=3, maximum_node_count=12, maximum_total_credentials=10)
    ar, dr, n = out["attacker_rewards"].cpu().numpy(), out["defender_rewards"].cpu().numpy(), out["lengths"].cpu().numpy()
    assert ar.shape == dr.shape and ar.shape[1] == 64 and ar.shape[0] == out["steps"] <= 60 and (n >= 1).all() and n.max() == out["steps"]
    assert (ar >= 0).all()                                  # modifier 0: an intercepted action costs nothing, penalties never surface (env.py:1169)
    out2 = simulate(25, "Random", "None", n_envs=8, seed=1, maximum_node_count=12, maximum_total_credentials=10, attacker_action_masking=True)
    assert out2["defender_rewards"] is None and out2["attacker_rewards"].shape == (25, 8) and float(out2["attacker_rewards"].sum()) > 0
