"""CPU tests of the host-side logic that sits above the C ABI — the SB3 VecEnv adapter (marlon_amd/vecenv.py) and the batched episode
loop (marlon_amd/simulate.py) — with scripted stand-ins for the GPU wrappers (torch CPU tensors), so that the conversions and the control
flow are checked here too; the same code against the HIP engine and the reference's traces: tests/test_gpu_vecenv.py, test_gpu_episodes.py.

What is pinned: DummyVecEnv / VecMonitor conventions (4-tuple, float32 rewards, bool dones, `TimeLimit.truncated = truncated and not
terminated`, `terminal_observation` and `episode{r,l,t}` only on done, baseline_marlon_agent.py:100-167) and marl_algorithm.run_episode's
rules (attacker then defender, stop on either done, max_steps, the defender's `-last attacker reward` after an attacker done)."""
import os
import types

import numpy as np
import pytest
import torch


class ScriptedAttacker:
    """Stands in for AttackerVecEnv: E envs, outputs read from a script [T, E]."""

    def __init__(self, rewards, terminated, truncated, auto_reset=True):
        self.torch = torch
        self.engine = types.SimpleNamespace(device=torch.device("cpu"))
        self.script = (torch.as_tensor(rewards, dtype=torch.float32), torch.as_tensor(terminated, dtype=torch.uint8),
                       torch.as_tensor(truncated, dtype=torch.uint8))
        self.num_envs = self.script[0].shape[1]
        self.auto_reset = auto_reset
        self.discrete, self.discrete_n, self.nvec = True, 7, np.array([3, 2, 2, 2, 2, 2, 2, 2, 2, 2])
        self.max_timesteps = 9
        self.t = 0
        self.resets = 0
        self.ret = torch.zeros(self.num_envs, dtype=torch.float64)
        self.len = torch.zeros(self.num_envs, dtype=torch.int32)
        self.actions_seen = []
        self._obs = {"discovered_node_count": torch.zeros(self.num_envs, dtype=torch.int32), "connect": torch.zeros((self.num_envs, 2, 2), dtype=torch.int8)}
        self._terminal = {k: torch.zeros_like(v) for k, v in self._obs.items()}

    @property
    def observation(self):
        return self._obs

    @property
    def terminal_observation(self):
        return self._terminal

    def action_masks(self):
        m = torch.zeros((self.num_envs, self.discrete_n), dtype=torch.bool)
        m[:, self.t % self.discrete_n] = True
        return m

    def reset(self):
        self.resets += 1
        self.ret.zero_()
        self.len.zero_()
        for v in self._obs.values():
            v.zero_()
        return self._obs

    def step(self, actions):
        self.actions_seen.append(np.asarray(actions).copy())
        r, te, tr = (x[self.t] for x in self.script)
        self.t += 1
        self._obs["discovered_node_count"] += 1
        self._obs["connect"] += 1
        self.ret += r.double()
        self.len += 1
        info = {"invalid_action": torch.zeros(self.num_envs, dtype=torch.bool), "cyber_step_executed": torch.ones(self.num_envs, dtype=torch.bool),
                "network_availability": torch.ones(self.num_envs, dtype=torch.float64), "step_count": self.len.clone(),
                "episode_return": self.ret.clone(), "episode_length": self.len.clone()}
        done = (te | tr) != 0
        if self.auto_reset and bool(done.any()):
            for k in self._obs:
                self._terminal[k][done] = self._obs[k][done]
                self._obs[k][done] = 0
            self.ret[done] = 0
            self.len[done] = 0
        return self._obs, r.clone(), te.clone(), tr.clone(), info

    def close(self):
        pass


def test_vecenv_adapter_follows_dummyvecenv_and_vecmonitor_conventions():
    from marlon_amd.vecenv import MarlonVecEnv
    rewards = [[1, 0, 2], [3, 5, 0], [0, 7, 1], [2, 0, 0]]
    terminated = [[0, 0, 0], [0, 1, 0], [0, 0, 0], [1, 0, 0]]
    truncated = [[0, 0, 0], [0, 0, 0], [0, 0, 1], [1, 0, 0]]
    env = MarlonVecEnv(ScriptedAttacker(rewards, terminated, truncated))
    assert env.num_envs == 3 and env.observation_space is None          # no gymnasium in this image (and the scripted stand-in has no topology)
    obs = env.reset()
    assert isinstance(obs["connect"], np.ndarray) and obs["connect"].shape == (3, 2, 2)
    masks = np.stack(env.env_method("action_masks"))                    # sb3_contrib get_action_masks
    assert masks.shape == (3, 7) and masks.dtype == np.bool_ and masks[:, 0].all() and masks.sum() == 3
    ret, length = np.zeros(3), np.zeros(3, int)
    for t in range(4):
        env.step_async(np.array([t, t, t]))
        obs, r, dones, infos = env.step_wait()
        assert r.dtype == np.float32 and dones.dtype == np.bool_ and isinstance(infos, list) and len(infos) == 3
        np.testing.assert_array_equal(r, np.array(rewards[t], np.float32))
        np.testing.assert_array_equal(dones, np.array(terminated[t], bool) | np.array(truncated[t], bool))
        ret += rewards[t]
        length += 1
        for i in range(3):
            assert infos[i]["TimeLimit.truncated"] == bool(truncated[t][i] and not terminated[t][i])
            if dones[i]:
                assert infos[i]["episode"]["r"] == ret[i] and infos[i]["episode"]["l"] == length[i] and infos[i]["episode"]["t"] >= 0
                assert int(infos[i]["terminal_observation"]["discovered_node_count"]) == length[i]      # the episode's LAST observation
                assert obs["discovered_node_count"][i] == 0                                             # the returned one is the reset observation
                ret[i], length[i] = 0.0, 0
            else:
                assert "episode" not in infos[i] and "terminal_observation" not in infos[i]
                assert obs["discovered_node_count"][i] == length[i]
    with pytest.raises(RuntimeError, match="step_wait"):
        env.step_wait()
    assert env.get_attr("max_timesteps") == [9, 9, 9] and env.get_attr("len", indices=[1]) == [2]
    assert env.env_is_wrapped(object) == [False] * 3 and env.seed(5) == [5, 6, 7]
    with pytest.raises(ValueError, match="auto_reset=True"):
        MarlonVecEnv(ScriptedAttacker(rewards, terminated, truncated, auto_reset=False))
    dev = MarlonVecEnv(ScriptedAttacker(rewards, terminated, truncated), numpy_outputs=False)
    dev.reset()
    o, r, d, infos = dev.step(np.zeros(3))
    assert isinstance(r, torch.Tensor) and isinstance(d, torch.Tensor) and d.dtype == torch.bool and isinstance(o["connect"], torch.Tensor)


class ScriptedDefender:
    def __init__(self, attacker, rewards, terminated, truncated):
        self.attacker, self.torch, self.engine, self.num_envs = attacker, torch, attacker.engine, attacker.num_envs
        self.script = (torch.as_tensor(rewards, dtype=torch.float64), torch.as_tensor(terminated, dtype=torch.uint8),
                       torch.as_tensor(truncated, dtype=torch.uint8))
        self.nvec = np.array([5, 3, 3, 6, 2, 3, 6, 2, 3, 3, 3, 3])
        self.t = 0
        self.kinds = []

    def reset(self, env_mask=None):
        return {}

    def step(self, actions):
        self.kinds.append(actions[:, 0].clone())
        r, te, tr = (x[self.t] for x in self.script)
        self.t += 1
        return {}, r.clone(), te.clone(), tr.clone(), {}


def test_run_episode_stop_rule_and_reset_request_rule():
    from marlon_amd.simulate import run_episode
    #            env 0: attacker done at step 2 | env 1: defender done at step 1 | env 2: runs to max_steps | env 3: both done at step 0
    a_r = [[1, 2, 3, 50], [4, 5, 6, 9], [7, 9, 1, 9], [9, 9, 2, 9]]
    a_te = [[0, 0, 0, 1], [0, 0, 0, 0], [1, 0, 0, 0], [0, 0, 0, 0]]
    a_tr = [[0] * 4] * 4
    d_r = [[-1, -2, -3, 77], [-4, -5000, -6, 77], [77, 77, -1, 77], [77, 77, -2, 77]]
    d_te = [[0, 0, 0, 1], [0, 1, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0]]
    d_tr = [[0] * 4] * 4
    att = ScriptedAttacker(a_r, a_te, a_tr, auto_reset=False)
    dfd = ScriptedDefender(att, d_r, d_te, d_tr)
    out = run_episode(att, dfd, lambda e: torch.zeros(4, dtype=torch.int64), lambda d: torch.zeros((4, 12), dtype=torch.int64), max_steps=4)
    assert att.resets == 1 and out["steps"] == 4
    np.testing.assert_array_equal(out["lengths"].numpy(), [3, 2, 4, 1])
    np.testing.assert_array_equal(out["attacker_done"].numpy(), [True, False, False, True])
    np.testing.assert_array_equal(out["defender_done"].numpy(), [True, True, False, True])
    np.testing.assert_array_equal(out["attacker_rewards"].numpy(), [[1, 2, 3, 50], [4, 5, 6, 0], [7, 0, 1, 0], [0, 0, 2, 0]])
    # the defender's reward after an ATTACKER done is -1 * the attacker's last reward (defend_wrapper.py:269-271), whatever the script says
    np.testing.assert_array_equal(out["defender_rewards"].numpy(), [[-1, -2, -3, -50], [-4, -5000, -6, 0], [-7, 0, -1, 0], [0, 0, -2, 0]])
    # envs whose episode is over, or whose attacker just ended it, take no defender turn on the device (kind -2)
    np.testing.assert_array_equal(torch.stack(dfd.kinds).numpy(), [[0, 0, 0, -2], [0, 0, 0, -2], [-2, -2, 0, -2], [-2, -2, 0, -2]])
    solo = run_episode(ScriptedAttacker(a_r, a_te, a_tr, auto_reset=False), None, lambda e: torch.zeros(4, dtype=torch.int64), max_steps=2)
    assert solo["defender_rewards"] is None and solo["steps"] == 2 and solo["lengths"].tolist() == [2, 2, 2, 1]
    with pytest.raises(ValueError, match="auto_reset=False"):
        run_episode(ScriptedAttacker(a_r, a_te, a_tr, auto_reset=True), None)


def test_no_unbound_names_in_host_code():
    """bench.py, tools/ and the package run for real only on the GPU box; a misspelt or missing name must not wait for it."""
    from tools import lint_names
    problems = [(f, lint_names.unbound(os.path.join(lint_names.REPO, f))) for f in lint_names.DEFAULT]
    assert not [p for p in problems if p[1]], [p for p in problems if p[1]]


def test_lazy_info_list_is_a_list_of_dicts_built_on_demand():
    """MarlonVecEnv's infos (marlon_amd/vecenv.py _InfoList): a real `list` whose per-env dicts — DummyVecEnv's keys, VecMonitor's
    `episode`, `terminal_observation` — only exist once somebody indexes or iterates it (SB3's _update_info_buffer does), built from
    the step's info block; dicts handed out before the bulk build stay the ones in the list."""
    import copy
    import numpy as np
    from marlon_amd.vecenv import MarlonVecEnv, _InfoList
    layout = MarlonVecEnv._INFO_LAYOUT
    cols = {"network_availability": np.array([1, .9, .8, 1, 1.]), "episode_return": np.array([0, 5, 0, 7, 0.]), "rewards": np.arange(5, dtype=np.float32),
            "step_count": np.arange(5, dtype=np.int32), "episode_length": np.array([0, 3, 0, 4, 0], np.int32), "invalid_action": np.array([0, 1, 0, 0, 1], np.uint8),
            "terminated": np.array([0, 1, 0, 0, 0], np.uint8), "truncated": np.array([0, 0, 0, 1, 0], np.uint8)}
    raw = np.concatenate([cols[k].astype(dt).view(np.uint8) for k, dt in layout])
    calls = []

    def terminal_rows(ended):
        calls.append(list(ended))
        return {"x": np.arange(len(ended)) * 10}

    infos = _InfoList(5, raw, None, layout, terminal_rows, True, 1.5, MarlonVecEnv._INFO_KEYS)
    assert isinstance(infos, list) and len(infos) == 5 and not calls and list.__len__(infos) == 0      # nothing built, nothing fetched
    d3 = infos[3]                                                                                        # one dict on demand
    assert d3 == {"invalid_action": False, "cyber_step_executed": True, "network_availability": 1.0, "step_count": 3, "TimeLimit.truncated": True,
                  "terminal_observation": {"x": 10}, "episode": {"r": 7.0, "l": 4, "t": 1.5}}
    assert calls == [[1, 3]] and list.__len__(infos) == 0
    d3["extra"] = 1
    assert [i.get("episode") for i in infos] == [None, {"r": 5.0, "l": 3, "t": 1.5}, None, {"r": 7.0, "l": 4, "t": 1.5}, None]   # SB3's loop
    assert list.__len__(infos) == 5 and infos[3] is d3 and infos[-1]["invalid_action"] is True and infos[1]["TimeLimit.truncated"] is False
    assert "terminal_observation" not in infos[0] and infos.ended().tolist() == [1, 3] and infos.columns()["step_count"].tolist() == [0, 1, 2, 3, 4]
    assert infos[:2] == [infos[0], infos[1]] and len(copy.copy(infos)) == 5 and infos == list(infos) and calls == [[1, 3]]
