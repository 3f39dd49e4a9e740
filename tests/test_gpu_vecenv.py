"""The SB3 VecEnv calling surface (marlon_amd/vecenv.py) driven with the exact call sequence of the reference's rollout step
(marlon/baseline_models/multiagent/baseline_marlon_agent.py:100-167: get_action_masks(env) -> env_method("action_masks"),
env.step(actions) -> 4-tuple, _update_info_buffer(infos)) and checked against the traces captured from the reference's own
AttackerEnvWrapper / MaskedDiscreteAttackerWrapper (tests/golden/wrap_*.npz): rewards, dones, TimeLimit.truncated, terminal
observations, reset observations, action masks, and VecMonitor's episode statistics."""
import json
import os
import zlib

import numpy as np
import pytest

from tests import parity
from tests.test_gpu_facades import FLAT, WRAP

pytestmark = pytest.mark.gpu


def _flat_equal(obs, i, z, prefix, t, ctx):
    from marlon_amd.cyberbattle_env import SCALAR_KEYS
    ref = (lambda k: z[prefix + k]) if t is None else (lambda k: z[prefix + k][t])
    assert [int(np.asarray(obs[k]).reshape(-1)[i] if np.asarray(obs[k]).ndim == 1 else np.asarray(obs[k])[i]) for k in SCALAR_KEYS] == ref("scalars").tolist(), ctx + " scalars"
    for k in FLAT:
        np.testing.assert_array_equal(np.asarray(obs[k][i]).reshape(-1), np.asarray(ref(k)).reshape(-1), err_msg=f"{ctx} {k}")


def _make(name, n_envs=1, **extra):
    from marlon_amd import cyberbattle_env as ce
    from marlon_amd._abi import RNG_TAPE
    from marlon_amd.vecenv import MarlonVecEnv
    from marlon_amd.wrappers import AttackerVecEnv
    z = np.load(os.path.join(parity.GOLDEN, name + ".npz"))
    sj = json.loads(bytes(z["spec_json"]).decode())
    topo = parity.topology_for(name[len("wrap_"):])
    d = sj["defender"]
    att = AttackerVecEnv(topo, n_envs, maximum_node_count=sj["maximum_node_count"], maximum_total_credentials=sj["maximum_total_credentials"],
                         attacker_goal=ce.AttackerGoal(**sj["attacker_goal"]), defender_constraint=ce.DefenderConstraint(sj["maintain_sla"]),
                         defender_agent=None if d is None else ce.ScanAndReimageCompromisedMachines(d[1], d[2], d[3]),
                         max_timesteps=sj["max_timesteps"], discrete=sj["discrete"], rng_kind=RNG_TAPE, **extra)
    return z, sj, MarlonVecEnv(att)


@pytest.mark.parametrize("name", WRAP)
def test_vecenv_call_sequence_reproduces_reference_wrapper_traces(name):
    z, sj, env = _make(name)
    assert env.num_envs == 1
    last_obs = env.reset()                                                     # OnPolicyAlgorithm._setup_learn: self._last_obs = env.reset()
    _flat_equal(last_obs, 0, z, "first_", None, name + " reset")
    ep_info_buffer, resets, ret, length, prev_mask_ok = [], 0, 0.0, 0, False
    for t in range(len(z["reward"])):
        ctx = f"{name} step {t}"
        action_masks = np.stack(env.env_method("action_masks"))                # sb3_contrib.common.maskable.utils.get_action_masks
        assert action_masks.shape == (1, env.venv.discrete_n) and action_masks.dtype == np.bool_
        if prev_mask_ok:                                                       # the masks the policy sees now = the ones recorded after step t-1
            assert int(action_masks.sum()) == z["mask_sum"][t - 1] and zlib.crc32(action_masks[0].astype(np.int8).tobytes()) == z["mask_crc"][t - 1], ctx + " masks"
        if z["tape"].size:
            env.venv.engine.set_draw_tape(z["tape"][t:t + 1])
        clipped_actions = np.asarray(z["action"][t]).reshape((1,) if sj["discrete"] else (1, 10))
        step_result = env.step(clipped_actions)
        assert len(step_result) == 4                                            # VecEnv API: (obs, rewards, dones, infos)
        new_obs, rewards, dones, infos = step_result
        assert isinstance(infos, list) and len(infos) == 1 and isinstance(infos[0], dict)
        assert rewards.dtype == np.float32 and dones.dtype == np.bool_ and rewards.shape == (1,) and dones.shape == (1,)
        assert float(rewards[0]) == z["reward"][t], f"{ctx} reward {rewards[0]} != {z['reward'][t]}"
        term, trunc = bool(z["terminated"][t]), bool(z["truncated"][t])
        assert bool(dones[0]) == (term or trunc), ctx + " done"
        assert infos[0]["TimeLimit.truncated"] == (trunc and not term), ctx + " TimeLimit.truncated"
        assert infos[0]["invalid_action"] == bool(z["invalid"][t]) and infos[0]["cyber_step_executed"] == (not z["invalid"][t]), ctx
        ret += float(z["reward"][t])
        length += 1
        for info in infos:                                                      # BaseAlgorithm._update_info_buffer
            maybe_ep_info = info.get("episode")
            if maybe_ep_info is not None:
                ep_info_buffer.append(maybe_ep_info)
        if dones[0]:
            assert z["was_reset"][t]
            _flat_equal({k: v[np.newaxis] for k, v in infos[0]["terminal_observation"].items()}, 0, z, "", t, ctx + " terminal observation")
            _flat_equal(new_obs, 0, z, "after_reset_", resets, ctx + " observation after the auto-reset")
            assert ep_info_buffer[-1]["r"] == ret and ep_info_buffer[-1]["l"] == length and ep_info_buffer[-1]["t"] >= 0.0, ctx + " episode statistics"
            ret, length, resets = 0.0, 0, resets + 1
            prev_mask_ok = False
        else:
            assert "terminal_observation" not in infos[0] and "episode" not in infos[0]
            _flat_equal(new_obs, 0, z, "", t, ctx)
            prev_mask_ok = True
        last_obs = new_obs
    assert len(ep_info_buffer) == int(z["was_reset"].sum()) > 0
    assert env.get_attr("max_timesteps") == [sj["max_timesteps"]] and env.env_is_wrapped(object) == [False]
    assert env.get_attr("timesteps") == [length]
    env.close()


@pytest.mark.parametrize("name", WRAP)
def test_one_launch_wrapper_step_reproduces_reference_wrapper_traces(name):
    """The traces captured from the reference's own AttackerEnvWrapper / MaskedDiscreteAttackerWrapper, replayed through the wrapper
    WITHOUT materialised masks — for these small topologies the whole step is then ONE launch (mcbs_wrapper_fused.hip): rewards, flags,
    every non-mask observation field, terminal and reset observations, and the action mask itself, recovered from mcbs_mask_logits on the
    digest the launch left (attack_wrapper.py:255-372, action_masking.py:90-142)."""
    import torch
    z, sj, env = _make(name, materialize_masks=False)
    assert env.venv.engine.wrapper_step_launches(False) == 1
    nomask = [k for k in FLAT if k not in ("local_vulnerability", "remote_vulnerability", "connect")]

    def small_equal(obs, ref_prefix, t, ctx):
        from marlon_amd.cyberbattle_env import SCALAR_KEYS
        ref = (lambda k: z[ref_prefix + k]) if t is None else (lambda k: z[ref_prefix + k][t])
        assert [int(np.asarray(obs[k]).reshape(-1)[0]) for k in SCALAR_KEYS] == ref("scalars").tolist(), ctx + " scalars"
        for k in nomask:
            np.testing.assert_array_equal(np.asarray(obs[k][0]).reshape(-1), np.asarray(ref(k)).reshape(-1), err_msg=f"{ctx} {k}")

    small_equal(env.reset(), "first_", None, name + " reset")
    A = env.venv.discrete_n
    resets = 0
    for t in range(len(z["reward"])):
        ctx = f"{name} step {t} (one launch)"
        if z["tape"].size:
            env.venv.engine.set_draw_tape(z["tape"][t:t + 1])
        new_obs, rewards, dones, infos = env.step(np.asarray(z["action"][t]).reshape((1,) if sj["discrete"] else (1, 10)))
        assert float(rewards[0]) == z["reward"][t], ctx + " reward"
        term, trunc = bool(z["terminated"][t]), bool(z["truncated"][t])
        assert bool(dones[0]) == (term or trunc) and infos[0]["TimeLimit.truncated"] == (trunc and not term), ctx + " flags"
        assert infos[0]["invalid_action"] == bool(z["invalid"][t]), ctx + " interception"
        if dones[0]:
            small_equal({k: v[np.newaxis] for k, v in infos[0]["terminal_observation"].items()}, "", t, ctx + " terminal observation")
            small_equal(new_obs, "after_reset_", resets, ctx + " observation after the auto-reset")
            resets += 1
        else:
            small_equal(new_obs, "", t, ctx)
            # the mask the reference's wrapper reported after this step, from the digest alone
            logits = torch.zeros((1, A), device=env.venv.engine.device)
            mask = (env.venv.mask_logits(logits, fill=1.0) == 0).cpu().numpy()[0]
            assert int(mask.sum()) == z["mask_sum"][t] and zlib.crc32(mask.astype(np.int8).tobytes()) == z["mask_crc"][t], ctx + " mask from the digest"
    assert resets == int(z["was_reset"].sum()) > 0
    env.close()


def test_one_launch_wrapper_step_at_the_headline_batch_size(monkeypatch):
    """65 536 Chain-10 envs (the batch bench.py's `wrapper` leg times): the one-launch step against round 2's three launches, actions a
    masked-greedy policy on random scores (masks from mcbs_mask_logits on the digests) with a share of intercepted ones, truncation at 30
    steps so that thousands of envs end and are re-initialised inside the launches: every output, observation, terminal observation,
    counter and the engines' canonical state at the end."""
    import torch
    from marlon_amd.samples import chainpattern
    from marlon_amd.wrappers import AttackerVecEnv
    from tests.test_gpu_parity import _compare_states
    E = 65536
    kw = dict(maximum_node_count=12, maximum_total_credentials=12, discrete=True, max_timesteps=30, materialize_masks=False)
    fused = AttackerVecEnv(chainpattern.new_environment(10), E, **kw)
    monkeypatch.setenv("MCBS_NO_FUSED_WRAPPER", "1")
    three = AttackerVecEnv(chainpattern.new_environment(10), E, **kw)
    monkeypatch.delenv("MCBS_NO_FUSED_WRAPPER")
    assert fused.engine.wrapper_step_launches(False) == 1 and three.engine.wrapper_step_launches(False) == 3
    dev = fused.engine.device
    g = torch.Generator(device=dev).manual_seed(11)
    ended = 0
    for t in range(70):
        scores = torch.rand((E, fused.discrete_n), generator=g, device=dev)
        a = three.mask_logits(scores, fill=-1.0).argmax(dim=1)                   # uniform over the valid actions of each env
        a[t % 13::13] = fused.discrete_n - 1                                     # every 13th env: an undiscovered node index (intercepted)
        del scores
        o1, r1, te1, tr1, i1 = fused.step(a)
        o2, r2, te2, tr2, i2 = three.step(a)
        ctx = f"step {t}"
        assert torch.equal(r1, r2) and torch.equal(te1, te2) and torch.equal(tr1, tr2), ctx + " rewards / flags"
        for k in i1:
            assert torch.equal(i1[k], i2[k]), f"{ctx} info {k}"
        for k in o1:
            assert torch.equal(o1[k], o2[k]), f"{ctx} observation {k}"
        if t % 10 == 9:
            for k, x in fused.terminal_observation.items():
                assert torch.equal(x, three.terminal_observation[k]), f"{ctx} terminal observation {k}"
            for k in ("timesteps", "valid_action_count", "invalid_action_count", "episode_returns"):
                assert torch.equal(getattr(fused, k), getattr(three, k)), f"{ctx} wrapper counter {k}"
        ended += int((te1 | tr1).sum())
    assert ended >= 2 * E
    _compare_states(fused.engine.get_state(), three.engine.get_state(), "after 70 steps")
    fused.close()
    three.close()


def test_vecenv_batch_matches_the_wrapper_it_adapts():
    """2 048 envs, Discrete actions sampled from the masks obtained through env_method: the adapter's 4-tuple, per-env infos, episode
    statistics and terminal observations against the same AttackerVecEnv stepped directly, and tensors instead of arrays with
    numpy_outputs=False."""
    import torch
    from marlon_amd import cyberbattle_env as ce
    from marlon_amd.samples import chainpattern
    from marlon_amd.vecenv import MarlonVecEnv
    from marlon_amd.wrappers import AttackerVecEnv
    E = 2048

    def mk():
        return AttackerVecEnv(chainpattern.new_environment(4), E, maximum_node_count=6, maximum_total_credentials=6,
                              attacker_goal=ce.AttackerGoal(own_atleast_percent=1.0), max_timesteps=30, discrete=True)
    a, raw = MarlonVecEnv(mk()), mk()
    dev_env = MarlonVecEnv(mk(), numpy_outputs=False)
    obs = a.reset()
    raw.reset()
    dev_env.reset()
    g = torch.Generator(device="cpu").manual_seed(5)
    ret, length, episodes = np.zeros(E), np.zeros(E, np.int64), 0
    for t in range(90):
        masks = np.stack(a.env_method("action_masks"))
        assert masks.shape == (E, a.venv.discrete_n) and np.array_equal(masks, a.action_masks())
        scores = torch.rand(masks.shape, generator=g).numpy()
        actions = np.where(masks, scores, -1.0).argmax(axis=1)
        if t % 7 == 3:
            actions[::5] = a.venv.discrete_n - 1                     # an undiscovered node index: intercepted, reward -1, env not stepped
        new_obs, rewards, dones, infos = a.step(actions)
        o2, r2, term2, trunc2, info2 = raw.step(actions)
        o3, r3, d3, i3 = dev_env.step(torch.as_tensor(actions, device=dev_env.venv.engine.device))
        assert isinstance(r3, torch.Tensor) and r3.is_cuda and isinstance(o3["connect"], torch.Tensor)
        np.testing.assert_array_equal(rewards, r2.cpu().numpy())
        np.testing.assert_array_equal(r3.cpu().numpy(), rewards)
        np.testing.assert_array_equal(dones, ((term2 | trunc2) != 0).cpu().numpy())
        np.testing.assert_array_equal(d3.cpu().numpy(), dones)
        for k in new_obs:
            np.testing.assert_array_equal(new_obs[k], o2[k].cpu().numpy(), err_msg=f"step {t} obs {k}")
        ret += rewards
        length += 1
        for i in np.flatnonzero(dones):
            assert infos[i]["episode"]["r"] == ret[i] and infos[i]["episode"]["l"] == length[i]
            assert infos[i]["TimeLimit.truncated"] == bool(trunc2[i] and not term2[i])
            for k in ("connect", "discovered_nodes_properties", "nodes_privilegelevel"):
                np.testing.assert_array_equal(infos[i]["terminal_observation"][k], raw.terminal_observation[k][i].cpu().numpy())
            episodes += 1
        for i in np.flatnonzero(~dones)[:50]:
            assert "episode" not in infos[i] and "terminal_observation" not in infos[i]
        ret[dones], length[dones] = 0.0, 0
    assert episodes >= E                                                 # every env ended (win or truncation at 30) at least once
    assert a.get_attr("timesteps", indices=[0, 5]) == [int(length[0]), int(length[5])]
    a.close(); raw.close(); dev_env.close()


def test_graph_replayed_wrapper_step_equals_eager():
    """AttackerVecEnv(use_graph=True): the whole wrapper step as one hipGraph replay (no host round trip) returns what the eager
    wrapper returns — rewards, flags, infos, observations, terminal observations, episode statistics — over episodes that end and
    reset inside the graph; with and without materialised masks; with an in-env defender (Philox)."""
    import torch
    from marlon_amd import cyberbattle_env as ce
    from marlon_amd.wrappers import AttackerVecEnv
    E = 1024
    for lean in (False, True):
        kw = dict(maximum_node_count=12, maximum_total_credentials=10, attacker_goal=ce.AttackerGoal(own_atleast=6),
                  defender_agent=ce.ScanAndReimageCompromisedMachines(0.6, 2, 5), defender_constraint=ce.DefenderConstraint(0.8),
                  max_timesteps=20, discrete=True, seed=5, materialize_masks=not lean)
        eager = AttackerVecEnv(parity.topology_for("toyctf"), E, **kw)
        graph = AttackerVecEnv(parity.topology_for("toyctf"), E, use_graph=True, **kw)
        ref = eager if not lean else AttackerVecEnv(parity.topology_for("toyctf"), E, **dict(kw, materialize_masks=True))
        g = torch.Generator(device=eager.engine.device).manual_seed(11)
        ended = 0
        for t in range(70):
            m = ref.action_masks()
            scores = torch.rand(m.shape, generator=g, device=m.device)
            actions = torch.where(m, scores, torch.full_like(scores, -1.0)).argmax(dim=1)
            if t % 6 == 2:
                actions[::9] = eager.discrete_n - 1
            o1, r1, te1, tr1, i1 = eager.step(actions)
            o2, r2, te2, tr2, i2 = graph.step(actions)
            if lean:
                ref.step(actions)
            assert torch.equal(r1, r2) and torch.equal(te1, te2) and torch.equal(tr1, tr2), f"lean={lean} step {t}"
            for k in ("invalid_action", "network_availability", "step_count", "episode_return", "episode_length"):
                assert torch.equal(i1[k], i2[k]), f"lean={lean} step {t} info {k}"
            for k in o1:
                assert torch.equal(o1[k], o2[k]), f"lean={lean} step {t} obs {k}"
            d = ((te1 | tr1) != 0)
            if bool(d.any()):
                for k in eager.terminal_observation:
                    assert torch.equal(eager.terminal_observation[k][d], graph.terminal_observation[k][d]), f"lean={lean} step {t} terminal {k}"
                ended += int(d.sum())
        assert ended >= E
        eager.close(); graph.close()
        if lean:
            ref.close()
    with pytest.raises(ValueError, match="draw tape"):
        AttackerVecEnv(parity.topology_for("toyctf"), 4, maximum_node_count=12, maximum_total_credentials=10, use_graph=True, rng_kind=1)


def test_defender_vecenv_adapter_follows_the_reference_trace_to_its_first_episode_end():
    """DefenderVecEnvAdapter (the SB3 VecEnv surface over DefenderVecEnv) on the joint attacker / defender trace captured from
    marlon's DefenderEnvWrapper + LearningDefender: 4-tuple, float32 rewards, bool dones, per-env infos; up to the first episode end
    every reward / done / observation equals the reference's, and that step carries terminal_observation + episode statistics."""
    from marlon_amd import cyberbattle_env as ce
    from marlon_amd.vecenv import DefenderVecEnvAdapter
    from marlon_amd.wrappers import AttackerVecEnv, DefenderVecEnv
    from tests.test_gpu_facades import DEF_KEYS
    name = "wrap_defender_toyctf_s72"
    z = np.load(os.path.join(parity.GOLDEN, name + ".npz"))
    sj = json.loads(bytes(z["spec_json"]).decode())
    att = AttackerVecEnv(parity.topology_for("toyctf"), 1, maximum_node_count=12, maximum_total_credentials=10,
                         attacker_goal=ce.AttackerGoal(**sj["attacker_goal"]), defender_constraint=ce.DefenderConstraint(sj["maintain_sla"]),
                         losing_reward=sj["losing_reward"], max_timesteps=sj["max_timesteps"], auto_reset=False, learned_defender=True)
    env = DefenderVecEnvAdapter(DefenderVecEnv(att, max_timesteps=sj["max_timesteps"], invalid_action_reward=-1, loss_reward=-5000.0))
    obs = env.reset()
    for k in DEF_KEYS:
        np.testing.assert_array_equal(obs[k][0], z["first_" + k])
    ret, steps = 0.0, 0
    for t in range(len(z["a_reward"])):
        att.step(z["a_action"][t].reshape(1, 10))
        if z["d_action"][t][0] <= -2:
            break                                        # the attacker ended the episode first in the trace: nothing more to compare
        step_result = env.step(z["d_action"][t].reshape(1, 12))
        assert len(step_result) == 4
        dobs, rewards, dones, infos = step_result
        assert rewards.dtype == np.float32 and dones.dtype == np.bool_ and isinstance(infos[0], dict)
        assert float(rewards[0]) == np.float32(z["d_reward"][t]), f"step {t} reward"
        done = bool(z["d_terminated"][t] or z["d_truncated"][t])
        assert bool(dones[0]) == done and infos[0]["valid_action"] == bool(z["d_valid"][t]), f"step {t}"
        assert infos[0]["TimeLimit.truncated"] == bool(z["d_truncated"][t] and not z["d_terminated"][t])
        ret += float(z["d_reward"][t])
        steps += 1
        if done:
            for k in DEF_KEYS:
                np.testing.assert_array_equal(infos[0]["terminal_observation"][k], z["d_" + k][t], err_msg=f"step {t} terminal {k}")
            assert infos[0]["episode"]["l"] == steps and infos[0]["episode"]["r"] == ret
            break
        for k in DEF_KEYS:
            np.testing.assert_array_equal(dobs[k][0], z["d_" + k][t], err_msg=f"step {t} {k}")
    assert steps > 5
    assert env.get_attr("max_timesteps") == [sj["max_timesteps"]] and env.env_is_wrapped(object) == [False]
    att.close()


def test_observe_masked_touches_only_the_flagged_envs():
    """mcbs_observe_masked (the reset observation of the envs a VecEnv just reset; the mask is scanned 64 envs per wavefront): flagged
    envs get exactly what a full mcbs_observe writes, the others keep their bytes — fused masks (Chain-10) and the separate mask
    kernels of a larger action space (24 nodes)."""
    import torch
    from marlon_amd import engine
    from marlon_amd._abi import RNG_PHILOX
    for trace, E in (("chain10_mix_s3", 1000), ("random24_defender_s51", 200)):
        _, sj = parity.load_trace(trace)
        topo = parity.topology_for(trace)
        spec = parity.spec_from_json(sj, n_envs=E, auto_reset=True, rng_kind=RNG_PHILOX, seed=3, max_episode_steps=50)
        eng = engine.BatchEngine(topo, spec)
        for t in range(30):
            eng.step(eng.sample_actions(True, seed=2, step=t))
        fields = list(parity.OBS_FIELDS) + ["mask_discrete"]
        full = eng.observe(eng.alloc_obs(fields))
        g = torch.Generator(device="cpu").manual_seed(1)
        for density in (0.0, 0.02, 0.5, 1.0):
            mask = (torch.rand(E, generator=g) < density).to(torch.uint8).to(eng.device)
            part = {k: torch.full_like(v, 77) for k, v in full.items()}
            eng.observe(part, env_mask=mask)
            sel = mask.bool()
            for k in fields:
                assert torch.equal(part[k][sel], full[k][sel]), f"{trace} density {density} {k}: flagged envs"
                assert bool((part[k][~sel] == 77).all()), f"{trace} density {density} {k}: other envs were touched"
        eng.close()


def test_graph_replayed_defender_turn_equals_eager():
    """DefenderVecEnv(use_graph=True) next to AttackerVecEnv(use_graph=True): the joint attacker / defender step as two hipGraph replays
    returns what the eager wrappers return (rewards in fp64, flags, validity, availability bits, the four observation fields)."""
    import torch
    from marlon_amd import cyberbattle_env as ce
    from marlon_amd.simulate import random_defender_policy
    from marlon_amd.wrappers import AttackerVecEnv, DefenderVecEnv
    E = 512
    kw = dict(maximum_node_count=12, maximum_total_credentials=10, attacker_goal=ce.AttackerGoal(own_atleast=6),
              defender_constraint=ce.DefenderConstraint(0.6), losing_reward=-5000.0, max_timesteps=60, auto_reset=False, learned_defender=True)
    a1 = AttackerVecEnv(parity.topology_for("toyctf"), E, **kw)
    a2 = AttackerVecEnv(parity.topology_for("toyctf"), E, use_graph=True, **kw)
    d1 = DefenderVecEnv(a1, max_timesteps=60, invalid_action_reward=-1)
    d2 = DefenderVecEnv(a2, max_timesteps=60, invalid_action_reward=-1, use_graph=True)
    pol = random_defender_policy(4)
    g = torch.Generator(device=a1.engine.device).manual_seed(2)
    nvec = torch.as_tensor(a1.nvec, device=a1.engine.device, dtype=torch.float64)
    for t in range(45):
        act = (torch.rand((E, 10), generator=g, device=nvec.device, dtype=torch.float64) * nvec).long()
        act[:, [1, 3, 4, 6, 7]] = act[:, [1, 3, 4, 6, 7]] % a1.observation["discovered_node_count"].long().clamp(min=1).unsqueeze(1)
        _, r1, te1, tr1, _ = a1.step(act)
        _, r2, te2, tr2, _ = a2.step(act)
        assert torch.equal(r1, r2) and torch.equal(te1, te2) and torch.equal(tr1, tr2), f"attacker step {t}"
        da = pol(d1)
        o1, q1, x1, y1, i1 = d1.step(da)
        o2, q2, x2, y2, i2 = d2.step(da)
        assert torch.equal(q1.view(torch.int64), q2.view(torch.int64)) and torch.equal(x1, x2) and torch.equal(y1, y2), f"defender step {t}"
        for k in ("valid_action", "network_availability", "sla_breached", "defender_won"):
            assert torch.equal(i1[k], i2[k]), f"defender step {t} info {k}"
        for k in o1:
            assert torch.equal(o1[k], o2[k]), f"defender step {t} obs {k}"
    assert bool((i1["network_availability"] < 1.0).any())            # re-imaging happened: the turns were not trivial
    a1.close(); a2.close()


@pytest.mark.parametrize("materialize_masks", [False, True])
def test_wrapper_finish_equals_the_separate_launches_it_replaces(materialize_masks):
    """mcbs_attacker_wrapper_finish (bookkeeping + terminal observation + reset + reset observation + cleared counters in one launch)
    against the five launches it replaces, which stay exported: mcbs_attacker_wrapper_post, mcbs_copy_rows_masked, mcbs_reset(mask),
    mcbs_observe_masked, mcbs_attacker_wrapper_clear — same actions, short episodes (truncation at 9 steps), every buffer compared
    after every step, the digest through mcbs_mask_logits."""
    import torch
    from marlon_amd import cyberbattle_env as ce
    from marlon_amd._abi import WrapperBuffers
    from marlon_amd.samples import toy_ctf
    from marlon_amd.wrappers import AttackerVecEnv
    E = 1024

    def mk():
        return AttackerVecEnv(toy_ctf.new_environment(), E, maximum_node_count=12, maximum_total_credentials=10,
                              attacker_goal=ce.AttackerGoal(own_atleast=3), max_timesteps=9, discrete=True, materialize_masks=materialize_masks,
                              defender_agent=ce.ScanAndReimageCompromisedMachines(0.6, 2, 5), seed=11)
    new, old = mk(), mk()
    n_done = torch.zeros(1, dtype=torch.int32, device=old.engine.device)

    def old_step(actions):                                  # round 2's _step_device, launch for launch
        eng = old.engine
        eng.decode_attacker_actions(discrete=actions, actions_out=old._rows, invalid_out=old._invalid)
        reward, terminated = eng.step_observe(old._rows, old._obs)
        wb = WrapperBuffers(*[x.data_ptr() for x in (
            old._invalid, reward, terminated, old.timesteps, old.valid_action_count, old.invalid_action_count, old.episode_returns,
            old.last_cyber_reward, old.has_cyber_reward, old._rewards, old._truncated, old._dones, old._ret_out, old._len_out, n_done)], None)
        eng.wrapper_post(wb, old.invalid_action_reward_modifier, old.max_timesteps)
        eng.copy_rows_masked([(old._obs[k], old._terminal[k]) for k in old._obs], old._dones)
        eng.reset(old._dones)
        eng.observe(old._obs, env_mask=old._dones)
        eng.wrapper_clear(wb)
    g = torch.Generator(device=new.engine.device).manual_seed(3)
    ended = 0
    for t in range(40):
        m = new.action_masks() if materialize_masks else None
        if m is None:                                       # sample from the masks the digest stands for
            logits = torch.rand((E, new.discrete_n), generator=g, device=new.engine.device)
            actions = new.mask_logits(logits, fill=-1.0).argmax(dim=1)
        else:
            actions = torch.where(m, torch.rand(m.shape, generator=g, device=m.device), torch.full((1,), -1.0, device=m.device)).argmax(dim=1)
        if t % 6 == 5:
            actions[::7] = new.discrete_n - 1               # intercepted actions
        new.step(actions)
        old_step(actions)
        for name in ("_rewards", "_truncated", "_dones", "_ret_out", "_len_out", "_invalid", "timesteps", "valid_action_count", "invalid_action_count",
                     "episode_returns", "last_cyber_reward", "has_cyber_reward"):
            assert torch.equal(getattr(new, name), getattr(old, name)), f"step {t}: {name}"
        assert torch.equal(new._executed, old._invalid == 0), f"step {t}: executed"
        assert int(n_done) == int(new._dones.sum())
        ended += int(n_done)
        for k in new._obs:
            assert torch.equal(new._obs[k], old._obs[k]), f"step {t}: observation {k}"
            assert torch.equal(new._terminal[k], old._terminal[k]), f"step {t}: terminal observation {k}"
        probe = torch.rand((E, new.discrete_n), generator=g, device=new.engine.device)
        assert torch.equal(new.mask_logits(probe.clone(), fill=-1.0), old.mask_logits(probe.clone(), fill=-1.0)), f"step {t}: digest"
        for x, y in zip(new.engine.get_state(), old.engine.get_state()):
            np.testing.assert_array_equal(x, y, err_msg=f"step {t}: environment state")
    assert ended >= 3 * E                                   # every env was reset inside the fused launch several times
    new.close(); old.close()


def test_wrapper_step_with_the_staged_hot_image_equals_default(monkeypatch):
    """mcbs_attacker_wrapper_step under MCBS_LDS_TOPO=1 (developer switch: hot image staged in LDS, the decode and the finish stay separate
    launches) returns what the default three-launch step returns, auto-resets included."""
    import torch
    from marlon_amd import cyberbattle_env as ce
    from marlon_amd.samples import toy_ctf
    from marlon_amd.wrappers import AttackerVecEnv
    E = 512

    def mk():
        return AttackerVecEnv(toy_ctf.new_environment(), E, maximum_node_count=12, maximum_total_credentials=10, attacker_goal=ce.AttackerGoal(own_atleast=3),
                              max_timesteps=8, discrete=True, defender_agent=ce.ScanAndReimageCompromisedMachines(0.6, 2, 5), seed=5)
    default = mk()
    monkeypatch.setenv("MCBS_LDS_TOPO", "1")
    staged = mk()
    monkeypatch.delenv("MCBS_LDS_TOPO")
    g = torch.Generator(device=default.engine.device).manual_seed(2)
    for t in range(30):
        m = default.action_masks()
        assert torch.equal(m, staged.action_masks())
        actions = torch.where(m, torch.rand(m.shape, generator=g, device=m.device), torch.full((1,), -1.0, device=m.device)).argmax(dim=1)
        if t % 5 == 4:
            actions[::3] = default.discrete_n - 1
        a, b = default.step(actions), staged.step(actions)
        for x, y in zip(a[1:4], b[1:4]):
            assert torch.equal(x, y), f"step {t}"
        for k in a[0]:
            assert torch.equal(a[0][k], b[0][k]), f"step {t}: observation {k}"
        for k in a[4]:
            assert torch.equal(a[4][k], b[4][k]), f"step {t}: info {k}"
        for k in default.terminal_observation:
            assert torch.equal(default.terminal_observation[k], staged.terminal_observation[k]), f"step {t}: terminal observation {k}"
    default.close(); staged.close()


@pytest.mark.parametrize("case", ["chain10_discrete", "toyctf_defender_multidiscrete", "chain4_discrete_graph", "toyctf_learned_defender"])
def test_one_launch_wrapper_step_equals_three_launches(case, monkeypatch):
    """mcbs_attacker_wrapper_step for small topologies without mask fields is ONE launch (mcbs_wrapper_fused.hip: decode, attacker's
    action, observation assembled in LDS before the defender's turn, defender / goals, bookkeeping, auto-reset, observation streamed
    out by the wavefront); MCBS_NO_FUSED_WRAPPER=1 keeps round 2's three launches.  Both wrappers take the same masked-random actions
    (intercepted ones included) through episode ends: every output, the observation and terminal observation, mcbs_mask_logits (the
    per-env digests) at every step and the engines' canonical state at intervals must agree.  Row shapes cover whole-vector rows
    (Chain-10 @12/12), rows that are not (Chain-4 @6/6: 6 privilege dwords), the in-env and the learned defender, hipGraph replay.
    attack_wrapper.py:255-372, action_masking.py:112-142, env.py:1153 vs 1156-1158."""
    import torch
    from marlon_amd import cyberbattle_env as ce
    from marlon_amd.samples import chainpattern, toy_ctf
    from marlon_amd.wrappers import AttackerVecEnv
    from tests.test_gpu_parity import _compare_states
    E = 1000                                                   # the last wavefront is partly empty
    if case == "chain10_discrete":
        env0, kw = chainpattern.new_environment(10), dict(maximum_node_count=12, maximum_total_credentials=12, discrete=True, max_timesteps=40)
    elif case == "toyctf_defender_multidiscrete":
        env0, kw = toy_ctf.new_environment(), dict(maximum_node_count=12, maximum_total_credentials=10, discrete=False, max_timesteps=35,
                                                    attacker_goal=ce.AttackerGoal(own_atleast=6, own_atleast_percent=1.0),
                                                    defender_constraint=ce.DefenderConstraint(0.8),
                                                    defender_agent=ce.ScanAndReimageCompromisedMachines(0.6, 2, 5), seed=31)
    elif case == "chain4_discrete_graph":
        env0, kw = chainpattern.new_environment(4), dict(maximum_node_count=6, maximum_total_credentials=6, discrete=True, max_timesteps=25, use_graph=True)
    else:
        env0, kw = toy_ctf.new_environment(), dict(maximum_node_count=12, maximum_total_credentials=10, discrete=True, max_timesteps=30, learned_defender=True)
    ref = AttackerVecEnv(env0, E, **{**kw, "use_graph": False})                         # masks materialised: the policy's masks come from here
    fused = AttackerVecEnv(env0, E, materialize_masks=False, **kw)
    monkeypatch.setenv("MCBS_NO_FUSED_WRAPPER", "1")
    three = AttackerVecEnv(env0, E, materialize_masks=False, **kw)
    monkeypatch.delenv("MCBS_NO_FUSED_WRAPPER")
    dev = ref.engine.device
    g = torch.Generator(device=dev).manual_seed(5)
    ended = 0
    for t in range(90):
        m = ref.action_masks()
        scores = torch.rand(m.shape, generator=g, device=dev)
        a = torch.where(m, scores, torch.full_like(scores, -1.0)).argmax(dim=1)
        if t % 7 == 3 or (t + 1) % kw["max_timesteps"] == 0:    # (also on the step many envs are truncated at: the terminal observation of an
            a[::9] = ref.discrete_n - 1                         # intercepted env is the one that STANDS, taken before the defender's last turn)
                                                                # undiscovered node indices: intercepted, the env keeps its observation
        if not kw["discrete"]:                                   # the same action as a MultiDiscrete(10) row (attack_wrapper.py:206-227)
            N, Cm = kw["maximum_node_count"], kw["maximum_total_credentials"]
            M, ML = ref._mask_split[0], ref._mask_split[1]
            L, R, P = ML // N, (ref.discrete_n - M - ML) // (N * N), M // (N * N * Cm)
            md = torch.zeros((E, 10), dtype=torch.int64, device=dev)
            is_c, is_l = a < M, (a >= M) & (a < M + ML)
            is_r = ~is_c & ~is_l
            md[:, 0] = torch.where(is_c, 2, torch.where(is_l, 0, 1))
            rel = torch.where(is_c, a, torch.where(is_l, a - M, a - M - ML))
            md[:, 1] = torch.where(is_l, rel // L, 0); md[:, 2] = torch.where(is_l, rel % L, 0)
            md[:, 3] = torch.where(is_r, rel // (N * R), 0); md[:, 4] = torch.where(is_r, (rel // R) % N, 0); md[:, 5] = torch.where(is_r, rel % R, 0)
            md[:, 6] = torch.where(is_c, rel // (N * P * Cm), 0); md[:, 7] = torch.where(is_c, (rel // (P * Cm)) % N, 0)
            md[:, 8] = torch.where(is_c, (rel // Cm) % P, 0); md[:, 9] = torch.where(is_c, rel % Cm, 0)
            a = md
        outs = [env.step(a.clone()) for env in (fused, three)]
        r_ref = ref.step(a)                                      # the mask-writing wrapper advances with the same actions
        (o1, r1, te1, tr1, i1), (o2, r2, te2, tr2, i2) = outs
        ctx = f"{case} step {t}"
        assert torch.equal(r1, r2) and torch.equal(te1, te2) and torch.equal(tr1, tr2), ctx + " rewards / flags"
        assert torch.equal(r1, r_ref[1]) and torch.equal(te1, r_ref[2]) and torch.equal(tr1, r_ref[3]), ctx + " vs the mask-writing wrapper"
        for k in i1:
            assert torch.equal(i1[k], i2[k]), f"{ctx} info {k}"
        for k in o1:
            assert torch.equal(o1[k], o2[k]), f"{ctx} observation {k}"
            assert torch.equal(o1[k], r_ref[0][k]), f"{ctx} observation {k} vs the mask-writing wrapper"
        for k, x in fused.terminal_observation.items():
            assert torch.equal(x, three.terminal_observation[k]), f"{ctx} terminal observation {k}"
        for k in ("timesteps", "valid_action_count", "invalid_action_count", "episode_returns", "last_cyber_reward", "has_cyber_reward"):
            assert torch.equal(getattr(fused, k), getattr(three, k)), f"{ctx} wrapper counter {k}"
        assert torch.equal(fused._rows, three._rows), ctx + " decoded rows"
        logits = torch.rand((E, fused.discrete_n), generator=g, device=dev)
        assert torch.equal(fused.mask_logits(logits.clone(), -1.0), three.mask_logits(logits.clone(), -1.0)), ctx + " mask_logits (digests)"
        ended += int((te1 | tr1).sum())
        if t % 30 == 29:
            _compare_states(fused.engine.get_state(), three.engine.get_state(), ctx)
    assert ended > E // 2
    for env in (ref, fused, three):
        env.close()


@pytest.mark.parametrize("topology", ["chain4", "toyctf", "tiny"])
def test_one_launch_wrapper_step_over_observation_bounds(topology, monkeypatch):
    """The one-launch wrapper step streams the observation's rows as 16-byte vectors where a row is whole vectors and as dwords where it
    is not, from LDS records sized by the bounds.  The same topology under a spread of (maximum_node_count, maximum_total_credentials)
    — rows of 3 .. 16 privilege dwords, 1 .. 16 cache rows, property rows of every phase — against the three-launch step and the
    mask-writing wrapper: outputs, observation, terminal observation, counters, digests (attack_wrapper.py:255-372,474-522)."""
    import torch
    from marlon_amd.samples import chainpattern, tinytoy, toy_ctf
    from marlon_amd.wrappers import AttackerVecEnv
    make = {"chain4": lambda: chainpattern.new_environment(4), "toyctf": toy_ctf.new_environment, "tiny": tinytoy.new_environment}[topology]
    n, c = {"chain4": (6, 5), "toyctf": (10, 5), "tiny": (3, 1)}[topology]
    E, T = 203, 12
    tried = set()
    for nm, cm in ((n, c), (n + 1, c + 1), (n + 2, c + 4), (9, 7), (11, 3), (13, 11), (14, 6), (15, 13), (16, 16), (16, c), (n, 16)):
        nm, cm = max(nm, n), max(cm, c)
        if (nm, cm) in tried:
            continue
        tried.add((nm, cm))
        kw = dict(maximum_node_count=nm, maximum_total_credentials=cm, discrete=True, max_timesteps=T)
        ref = AttackerVecEnv(make(), E, **kw)
        fused = AttackerVecEnv(make(), E, materialize_masks=False, **kw)
        assert fused.engine.wrapper_step_launches(False) == 1, f"{topology} ({nm}, {cm}): not the one-launch step"
        monkeypatch.setenv("MCBS_NO_FUSED_WRAPPER", "1")
        three = AttackerVecEnv(make(), E, materialize_masks=False, **kw)
        monkeypatch.delenv("MCBS_NO_FUSED_WRAPPER")
        dev = ref.engine.device
        g = torch.Generator(device=dev).manual_seed(nm * 31 + cm)
        for t in range(30):
            m = ref.action_masks()
            scores = torch.rand(m.shape, generator=g, device=dev)
            a = torch.where(m, scores, torch.full_like(scores, -1.0)).argmax(dim=1)
            if t % 5 == 2 or (t + 1) % T == 0:
                a[::7] = ref.discrete_n - 1                      # intercepted (undiscovered node index) unless everything is discovered
            (o1, r1, te1, tr1, i1), (o2, r2, te2, tr2, i2) = [env.step(a.clone()) for env in (fused, three)]
            r_ref = ref.step(a)
            ctx = f"{topology} bounds ({nm}, {cm}) step {t}"
            assert torch.equal(r1, r2) and torch.equal(te1, te2) and torch.equal(tr1, tr2), ctx + " rewards / flags"
            assert torch.equal(r1, r_ref[1]) and torch.equal(te1, r_ref[2]) and torch.equal(tr1, r_ref[3]), ctx + " vs the mask-writing wrapper"
            for k in i1:
                assert torch.equal(i1[k], i2[k]), f"{ctx} info {k}"
            for k in o1:
                assert torch.equal(o1[k], o2[k]), f"{ctx} observation {k}"
                assert torch.equal(o1[k], r_ref[0][k]), f"{ctx} observation {k} vs the mask-writing wrapper"
            for k, x in fused.terminal_observation.items():
                assert torch.equal(x, three.terminal_observation[k]), f"{ctx} terminal observation {k}"
                assert torch.equal(x, ref.terminal_observation[k]), f"{ctx} terminal observation {k} vs the mask-writing wrapper"
            for k in ("timesteps", "valid_action_count", "invalid_action_count", "episode_returns", "last_cyber_reward", "has_cyber_reward"):
                assert torch.equal(getattr(fused, k), getattr(three, k)), f"{ctx} wrapper counter {k}"
            logits = torch.rand((E, fused.discrete_n), generator=g, device=dev)
            assert torch.equal(fused.mask_logits(logits.clone(), -1.0), three.mask_logits(logits.clone(), -1.0)), ctx + " mask_logits (digests)"
        for env in (ref, fused, three):
            env.close()
    assert len(tried) >= 8


def test_terminal_observation_of_an_intercepted_last_action_is_the_one_that_stands(monkeypatch):
    """An action with an undiscovered node index does not step the env and returns the observation the env already had
    (attack_wrapper.py:286-308) — also when that very step truncates the episode: the terminal observation is then that standing
    observation, taken BEFORE the defender's turn of the last executed step (env.py:1153 vs 1156-1158), not a fresh look at the state the
    defender has changed since.  A defender that re-images every infected node in every step makes the two differ (privilege levels of a
    node the attacker owned in its last executed step); the one-launch step and the three-launch step must both hand out the standing one."""
    import torch
    from marlon_amd import cyberbattle_env as ce
    from marlon_amd.samples import chainpattern
    from marlon_amd.wrappers import AttackerVecEnv
    E, T = 6000, 14
    kw = dict(maximum_node_count=12, maximum_total_credentials=12, discrete=True, max_timesteps=T, attacker_goal=ce.AttackerGoal(own_atleast_percent=1.0),
              defender_constraint=ce.DefenderConstraint(0.0), defender_agent=ce.ScanAndReimageCompromisedMachines(1.0, 12, 1), seed=5)
    ref = AttackerVecEnv(chainpattern.new_environment(10), E, **kw)
    fused = AttackerVecEnv(chainpattern.new_environment(10), E, materialize_masks=False, **kw)
    monkeypatch.setenv("MCBS_NO_FUSED_WRAPPER", "1")
    three = AttackerVecEnv(chainpattern.new_environment(10), E, materialize_masks=False, **kw)
    monkeypatch.delenv("MCBS_NO_FUSED_WRAPPER")
    assert fused.engine.wrapper_step_launches(False) == 1 and three.engine.wrapper_step_launches(False) == 3
    dev = ref.engine.device
    g = torch.Generator(device=dev).manual_seed(2)
    standing = fresh_look = None
    for t in range(T):
        if t < T - 1:
            m = ref.action_masks()
            scores = torch.rand(m.shape, generator=g, device=dev)
            a = torch.where(m, scores, torch.full_like(scores, -1.0)).argmax(dim=1)
        else:
            a = torch.full((E,), ref.discrete_n - 1, dtype=torch.int64, device=dev)      # intercepted in every env; the T-th wrapper step truncates
        outs = [env.step(a.clone()) for env in (ref, fused, three)]
        for k in outs[1][0]:
            assert torch.equal(outs[1][0][k], outs[2][0][k]) and torch.equal(outs[1][0][k], outs[0][0][k]), f"step {t} observation {k}"
        if t == T - 2:
            standing = {k: v.clone() for k, v in outs[1][0].items()}
            fresh_look = ref.engine.observe(ref.engine.alloc_obs(["nodes_privilegelevel"]))["nodes_privilegelevel"].clone()
    never_ended = (ref.timesteps == 0) & (outs[0][3] != 0)                       # truncated by this step (their counters were just cleared)
    assert bool(outs[1][4]["invalid_action"].all()) and int(never_ended.sum()) > E // 2
    # the scenario is real: for some of those envs the state has moved on since the standing observation (a node owned in the last
    # executed step has been re-imaged by the defender: a fresh look shows privilege 0 where the standing observation shows 1)
    moved_on = (standing["nodes_privilegelevel"] != fresh_look).any(dim=1) & never_ended
    assert int(moved_on.sum()) > 0
    for k, x in fused.terminal_observation.items():
        sel = never_ended
        assert torch.equal(x[sel], three.terminal_observation[k][sel]), f"terminal observation {k}: one launch vs three"
        assert torch.equal(x[sel], ref.terminal_observation[k][sel]), f"terminal observation {k}: vs the mask-writing wrapper"
        assert torch.equal(x[sel], standing[k][sel]), f"terminal observation {k} is not the observation that stood"
    for env in (ref, fused, three):
        env.close()


def test_wrapper_steps_on_a_side_stream_equal_the_default_stream():
    """Every library call is enqueued on torch's CURRENT stream of the batch's device, queried per call (engine._stream: torch's raw-stream
    binding).  The same masked-random trajectory on the default stream and inside `torch.cuda.stream(side)` — actions produced and results
    consumed on that same side stream, no synchronize in between — must agree step for step."""
    import torch
    from marlon_amd.samples import chainpattern
    from marlon_amd.wrappers import AttackerVecEnv
    E, T = 3000, 40
    kw = dict(maximum_node_count=12, maximum_total_credentials=12, discrete=True, max_timesteps=17, materialize_masks=False)
    a_env, b_env = AttackerVecEnv(chainpattern.new_environment(10), E, **kw), AttackerVecEnv(chainpattern.new_environment(10), E, **kw)
    dev = a_env.engine.device
    side = torch.cuda.Stream(device=dev)
    assert a_env.engine._stream() == torch.cuda.current_stream(dev).cuda_stream
    with torch.cuda.stream(side):
        assert b_env.engine._stream() == side.cuda_stream != torch.cuda.default_stream(dev).cuda_stream
    ga, gb = torch.Generator(device=dev).manual_seed(3), torch.Generator(device=dev).manual_seed(3)
    ra, rb = [], []
    for t in range(T):
        la = torch.rand((E, a_env.discrete_n), generator=ga, device=dev)
        acts = a_env.mask_logits(la, fill=-1.0).argmax(dim=1)
        o, r, te, tr, info = a_env.step(acts)
        ra.append((r.clone(), te.clone(), tr.clone(), o["discovered_node_count"].clone()))
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for t in range(T):
            lb = torch.rand((E, b_env.discrete_n), generator=gb, device=dev)
            acts = b_env.mask_logits(lb, fill=-1.0).argmax(dim=1)
            o, r, te, tr, info = b_env.step(acts)
            rb.append((r.clone(), te.clone(), tr.clone(), o["discovered_node_count"].clone()))
    side.synchronize()
    for t, (x, y) in enumerate(zip(ra, rb)):
        for u, v in zip(x, y):
            assert torch.equal(u, v), f"step {t}: side stream differs from the default stream"
    assert sum(int((x[1] | x[2]).sum()) for x in ra) > E
    a_env.close(); b_env.close()
