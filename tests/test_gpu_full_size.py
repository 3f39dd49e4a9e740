"""GPU parity at BASELINE.json's OWN batch sizes (the headline's 65 536 Chain-10 envs, configs 3, 4 and 5; one GPU's shard for the
8-GPU ones).

The launch shape of every kernel depends on the batch size (workgroup-size ladder, LDS budget of the hot image,
grid-stride of the mask kernels), so the shapes the benchmark runs must meet a checker themselves, not only their
reduced cousins in test_gpu_parity.py.  For each configuration the HIP engine runs the FULL shard with Philox
defender draws and device-sampled actions, and

  * three windows of 256 envs spread over the batch (first, an unaligned one around the middle, last) are replayed by
    the CPU oracle — an oracle batch whose `env_id_base` is the window's global id, so the defender streams match —
    and compared on every output of every step (reward, raw reward, terminated, truncated, out-of-bound, step count,
    availability as fp64 bits) plus the canonical state and the small observation fields at intervals;
  * the whole batch is checked for determinism (two runs), shard invariance (two half-size shards with
    `env_id_base`, as bench.py's ranks use) and state invariants that hold for any action sequence.

Rules pinned: __process_outcome / connect (actions.py:325-423,524-606), ScanAndReimage (defender.py:42-55),
on_attacker_step_taken + reimage_node (actions.py:700-746).
"""
import numpy as np
import pytest

from tests import parity
from tests.test_gpu_parity import _compare_states

pytestmark = pytest.mark.gpu

WINDOW = 256
SMALL_OBS = ["scalars", "leaked_credentials", "credential_cache_matrix", "discovered_nodes_properties", "nodes_privilegelevel", "mask_local"]


def _config(name):
    from marlon_amd import flatten as F, model
    from marlon_amd.samples import chainpattern, random_net, toy_ctf
    if name == "headline_chain10_65536":
        # BASELINE.json's metric shape: Chain size=10, 65 536 envs, attacker only (bounds 12 / 12, goal own 100 %); episodes truncated at
        # 100 steps so that envs end — and are re-initialised by the wave-cooperative reset tail — inside the launches
        # (env.py:1145-1185 step, 1187-1209 reset, DummyVecEnv's auto-reset)
        return F.flatten(chainpattern.new_environment(10)), 65536, dict(
            maximum_node_count=12, maximum_total_credentials=12, attacker_goal=dict(own_atleast_percent=1.0)), 230, 100
    if name == "config3_toyctf_16384":
        # BASELINE.json configs[2] / SURVEY 8(d): ToyCtf, ScanAndReimage(0.6, 2, 5), SLA 0.80, own_atleast 6, N 12, C 10
        return F.flatten(toy_ctf.new_environment()), 16384, dict(
            maximum_node_count=12, maximum_total_credentials=10, attacker_goal=dict(own_atleast=6, own_atleast_percent=1.0),
            maintain_sla=0.80, defender=("scan_and_reimage", 0.6, 2, 5)), 160, 100
    if name == "config4_chain100_8192":
        # configs[3]: Chain size=100 (N 102, C 102), attacker + ScanAndReimage, 65 536 envs sharded 8 x 8 192
        return F.flatten(chainpattern.new_environment(100)), 8192, dict(
            maximum_node_count=102, maximum_total_credentials=102, attacker_goal=dict(own_atleast_percent=1.0),
            defender=("scan_and_reimage", 0.6, 2, 5)), 160, 120
    if name == "config5_random256_16384":
        # configs[4]: the config-5 generator at the engine's node limit (u8 node ids 0..255, four full 64-bit set words),
        # 131 072 envs sharded 8 x 16 384
        topo = F.flatten(random_net.build(model, 256, 0))
        return topo, 16384, dict(
            maximum_node_count=256, maximum_total_credentials=max(256, len(topo.triples)), maximum_discoverable_credentials_per_action=8,
            attacker_goal=dict(own_atleast_percent=1.0), maintain_sla=0.5, defender=("scan_and_reimage", 0.5, 4, 4)), 140, 90
    raise KeyError(name)


CONFIGS = ["headline_chain10_65536", "config3_toyctf_16384", "config4_chain100_8192", "config5_random256_16384"]


def _spec(kw, n_envs, base, max_steps, seed=20260):
    from marlon_amd._abi import RNG_PHILOX, EnvSpec
    return EnvSpec(n_envs=n_envs, auto_reset=True, max_episode_steps=max_steps, rng_kind=RNG_PHILOX, seed=seed, env_id_base=base, **kw)


@pytest.mark.parametrize("name", CONFIGS)
def test_full_shard_against_oracle_windows(name):
    from marlon_amd import engine
    from oracle.oracle import Oracle
    topo, E, kw, steps, max_steps = _config(name)
    base = 3 * E                                     # this shard = rank 3 of bench.py's split
    eng = engine.BatchEngine(topo, _spec(kw, E, base, max_steps))
    offs = [0, E // 2 - 100, E - WINDOW]             # the middle window straddles workgroup boundaries at an odd offset
    orcs = [Oracle(topo, _spec(kw, WINDOW, base + o, max_steps)) for o in offs]
    obs = eng.alloc_obs(SMALL_OBS)
    ended = 0
    for t in range(steps):
        a = eng.sample_actions(t % 4 != 3, seed=11, step=t)           # three valid draws, then one uniform (invalid actions too)
        check_obs = t % 40 == 39
        if check_obs:
            r, d = eng.step_observe(a, obs)
        else:
            r, d = eng.step(a)
        an = a.cpu().numpy()
        rn, dn = r.double().cpu().numpy(), d.cpu().numpy()
        info = {k: v.cpu().numpy() for k, v in eng.info.items()}
        state = eng.get_state() if (t % 40 == 39 or t == steps - 1) else None
        for o, orc in zip(offs, orcs):
            w = slice(o, o + WINDOW)
            ctx = f"{name} step {t} window {o}"
            oo = orc.alloc_obs(SMALL_OBS) if check_obs else None
            ref = orc.step(an[w], obs=oo)
            np.testing.assert_array_equal(rn[w], ref["reward"], err_msg=ctx + " reward")
            np.testing.assert_array_equal(info["raw_reward"][w].astype(np.float64), ref["raw_reward"], err_msg=ctx + " raw reward")
            np.testing.assert_array_equal(dn[w], ref["terminated"], err_msg=ctx + " terminated")
            np.testing.assert_array_equal(info["truncated"][w], ref["truncated"], err_msg=ctx + " truncated")
            np.testing.assert_array_equal(info["out_of_bound"][w], ref["oob"], err_msg=ctx + " oob")
            np.testing.assert_array_equal(info["step_count"][w], ref["step_count"], err_msg=ctx + " step_count")
            np.testing.assert_array_equal(info["network_availability"][w].view(np.uint64), ref["availability"].view(np.uint64),
                                          err_msg=ctx + " availability bits")
            if check_obs:
                for f in SMALL_OBS:
                    np.testing.assert_array_equal(obs[f][w].cpu().numpy(), oo[f], err_msg=f"{ctx} obs {f}")
            if state is not None:
                _compare_states(tuple(x[w] for x in state), orc.get_state(), ctx)
        ended += int(dn.sum()) + int(info["truncated"].sum())
    assert ended > 0                                 # episodes ended and were re-initialised inside the launch
    eng.close()


@pytest.mark.parametrize("name", CONFIGS)
def test_full_shard_properties(name):
    """Whole batch at the configuration's own size: determinism, 2-shard invariance, state invariants."""
    from marlon_amd import engine
    topo, E, kw, steps, max_steps = _config(name)
    steps = min(steps, 120)

    def run(n, base):
        eng = engine.BatchEngine(topo, _spec(kw, n, base, max_steps, seed=77))
        t_ = eng.torch
        tot = t_.zeros(n, dtype=t_.float64, device=eng.device)
        ends = t_.zeros(n, dtype=t_.int64, device=eng.device)
        avail_min = t_.ones(n, dtype=t_.float64, device=eng.device)
        for t in range(steps):
            r, d = eng.step(eng.sample_actions(True, seed=5, step=t))
            tot += r.double()
            ends += d.long() + eng.info["truncated"].long()
            avail_min = t_.minimum(avail_min, eng.info["network_availability"])
        st = eng.get_state()
        eng.close()
        return tot.cpu().numpy(), ends.cpu().numpy(), avail_min.cpu().numpy(), st

    tot, ends, amin, st = run(E, 0)
    tot2, ends2, amin2, st2 = run(E, 0)
    np.testing.assert_array_equal(tot, tot2)
    np.testing.assert_array_equal(ends, ends2)
    _compare_states(st, st2, name + " determinism")
    a0, e0, m0, s0 = run(E // 2, 0)
    a1, e1, m1, s1 = run(E // 2, E // 2)
    np.testing.assert_array_equal(np.concatenate([a0, a1]), tot)
    np.testing.assert_array_equal(np.concatenate([e0, e1]), ends)
    np.testing.assert_array_equal(np.concatenate([m0, m1]).view(np.uint64), amin.view(np.uint64))
    _compare_states(tuple(np.concatenate([x, y]) for x, y in zip(s0, s1)), st, name + " shard invariance")
    hdr, nodes, order, cache = st
    N = topo.n_nodes
    assert (hdr["n_discovered"] == nodes["discovered"].sum(axis=1)).all()
    assert (nodes["installed"] <= nodes["discovered"]).all()                       # an owned node was discovered first
    assert (nodes["installed"] <= nodes["ever_owned"]).all()
    assert (nodes["privilege"][nodes["installed"] == 1] >= 1).all()
    assert (nodes["installed"][nodes["running"] == 0] == 0).all()                  # a node being re-imaged carries no agent
    assert (nodes["countdown"][nodes["running"] == 1] == 0).all()
    assert ((nodes["attacked_since"] & ~nodes["attacked_ever"]) == 0).all()
    assert (hdr["step_count"] <= max_steps).all() and (hdr["n_creds"] <= len(topo.triples)).all()
    disc_sorted = np.sort(np.where(order < N, order, 0xFFFF), axis=1)
    assert ((disc_sorted[:, 1:] != disc_sorted[:, :-1]) | (disc_sorted[:, 1:] == 0xFFFF)).all()   # discovery order has no duplicates
    assert (ends > 0).all()                                                        # every env ended (goal, SLA or truncation) at least once
    assert tot.sum() > 0 and ("defender" not in kw or (amin < 1.0).any())          # the defender re-imaged something somewhere


def test_chain100_full_shard_observation():
    """Config 4's shard with the WHOLE observation (70 GB of int8 masks at 8 192 envs: the connect mask goes through
    mask_connect_rows_kernel with its grid capped in y): small fields and mask_remote of three 256-env windows and the
    connect mask of 16 envs per window equal the oracle's (oracle batches keyed by the windows' global env ids)."""
    from marlon_amd import engine
    from oracle.oracle import Oracle
    topo, E, kw, _, max_steps = _config("config4_chain100_8192")
    eng = engine.BatchEngine(topo, _spec(kw, E, 0, max_steps))
    offs = [0, E // 2 - 100, E - WINDOW]
    orcs = [Oracle(topo, _spec(kw, WINDOW, o, max_steps)) for o in offs]
    orcs16 = [Oracle(topo, _spec(kw, 16, o, max_steps)) for o in offs]      # 8.5 MB of connect mask per env on the host
    fields = SMALL_OBS + ["mask_remote", "mask_connect"]
    obs = eng.alloc_obs(fields)
    T = 24                                           # (no episode ends this early, so the last observation is of a live env)
    for t in range(T):
        a = eng.sample_actions(True, seed=3, step=t)
        last = t == T - 1
        if last:
            eng.step_observe(a, obs)
        else:
            eng.step(a)
        an = a.cpu().numpy()
        for o, orc, o16 in zip(offs, orcs, orcs16):
            if not last:
                orc.step(an[o:o + WINDOW])
                o16.step(an[o:o + 16])
                continue
            oo = orc.alloc_obs(SMALL_OBS + ["mask_remote"])
            orc.step(an[o:o + WINDOW], obs=oo)
            for f in oo:
                np.testing.assert_array_equal(obs[f][o:o + WINDOW].cpu().numpy(), oo[f], err_msg=f"window {o} obs {f}")
            oc = o16.alloc_obs(["mask_connect"])
            o16.step(an[o:o + 16], obs=oc)
            np.testing.assert_array_equal(obs["mask_connect"][o:o + 16].cpu().numpy(), oc["mask_connect"], err_msg=f"window {o} mask_connect")
            assert oc["mask_connect"].sum() > 0
    eng.close()
