"""GPU parity tests (run on the MI355X box with `pytest -m gpu`): the HIP engine, called through the C ABI,
against (1) the golden traces captured from the imported reference, (2) the CPU oracle on identical seeded
action batches, and (3) size-independent properties at BASELINE.json's full batch size."""
import numpy as np
import pytest

from tests import parity

pytestmark = pytest.mark.gpu


def _engine():
    from marlon_amd import engine
    return engine


class EngineStepper:
    """Single-env adapter of BatchEngine for parity.replay (tape-driven defender)."""

    def __init__(self, topo, spec):
        self.eng = _engine().BatchEngine(topo, spec)
        self.torch = self.eng.torch

    def reset_observation(self, fields):
        obs = self.eng.alloc_obs(fields)
        self.eng.observe(obs)
        return {k: v.cpu().numpy() for k, v in obs.items()}

    def step(self, actions, tape, want_obs):
        if tape is not None:
            self.eng.set_draw_tape(tape)
        if want_obs:
            obs = self.eng.alloc_obs(want_obs)
            r, d = self.eng.step_observe(actions, obs)
        else:
            obs = None
            r, d = self.eng.step(actions)
        out = dict(reward=r.double().cpu().numpy(), terminated=d.cpu().numpy(),
                   step_count=self.eng.info["step_count"].cpu().numpy(),
                   availability=self.eng.info["network_availability"].cpu().numpy(),
                   raw_reward=self.eng.info["raw_reward"].double().cpu().numpy(),
                   oob=self.eng.info["out_of_bound"].cpu().numpy())
        out["obs"] = None if obs is None else {k: v.cpu().numpy() for k, v in obs.items()}
        _, _, order, cache = self.eng.get_state()
        out["order"], out["cache"] = order, cache
        return out


@pytest.mark.parametrize("name", parity.trace_names())
def test_engine_matches_reference_trace_with_observations(name):
    """mcbs_step_observe == the reference, step by step: rewards, flags, availability bits, all observation fields."""
    limit = 60 if name.startswith("chain100") else 0      # 8.5 MB connect mask per step
    assert parity.replay(name, EngineStepper, check_obs=True, limit=limit) > 0


@pytest.mark.parametrize("name", parity.trace_names())
def test_fused_step_matches_reference_trace(name):
    """mcbs_step (one fused launch, no observation) == the reference on rewards / termination / availability / order."""
    assert parity.replay(name, EngineStepper, check_obs=False) > 0


def _compare_states(a, b, ctx):
    for x, y, what in zip(a, b, ("header", "nodes", "order", "cache")):
        if x.dtype.names:
            for f in x.dtype.names:
                if f.startswith("pad"):
                    continue
                np.testing.assert_array_equal(x[f], y[f], err_msg=f"{ctx}: state {what}.{f}")
        else:
            np.testing.assert_array_equal(x, y, err_msg=f"{ctx}: state {what}")


CASES = {
    # name: (topology trace prefix, spec overrides, n_envs, steps)
    "chain10_attacker": ("chain10_script", dict(), 4096, 200),
    "toyctf_defender": ("toyctf_defender_s11", dict(), 2048, 200),
    "chain100_defender": ("chain100_defender_s31", dict(), 512, 120),
    "sink_defender": ("sink_defender_s43", dict(), 2048, 300),
    "sink_evict": ("sink_evict_s44", dict(), 1024, 200),
    "random24_defender": ("random24_defender_s51", dict(), 1024, 200),
    "ad6_wide_cache": ("ad6_mix_s70", dict(), 512, 200),          # 821 cacheable credentials: the set lives in memory / LDS
    "random_s5_defender": ("random_s5_defender_s67", dict(), 512, 150),
    # ExternalRandomEvents with Philox draws: per-env vulnerability keys, service bits and firewall lists diverge between envs
    "toyctf_randomevents": ("toyctf_randomevents_s81", dict(), 1024, 200),
    "sink_randomevents": ("sink_randomevents_s83", dict(), 1024, 200),
}


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("policy", ["valid", "uniform"])
def test_engine_matches_oracle_batched(case, policy):
    """Thousands of envs, Philox defender draws, device-sampled actions, auto-reset: every output and the full
    canonical state equal the CPU oracle's."""
    from marlon_amd._abi import RNG_PHILOX
    from oracle.oracle import Oracle
    trace, over, E, steps = CASES[case]
    _, sj = parity.load_trace(trace)
    topo = parity.topology_for(trace)
    spec = parity.spec_from_json(sj, n_envs=E, auto_reset=True, rng_kind=RNG_PHILOX, seed=0xC0FFEE + len(case),
                                 env_id_base=1000, max_episode_steps=150, **over)
    eng = _engine().BatchEngine(topo, spec)
    orc = Oracle(topo, spec)
    for t in range(steps):
        a = eng.sample_actions(policy == "valid", seed=77, step=t)
        r, d = eng.step(a)
        o = orc.step(a.cpu().numpy())
        ctx = f"{case}/{policy} step {t}"
        np.testing.assert_array_equal(r.double().cpu().numpy(), o["reward"], err_msg=ctx + " reward")
        np.testing.assert_array_equal(eng.info["raw_reward"].double().cpu().numpy(), o["raw_reward"], err_msg=ctx + " raw")
        np.testing.assert_array_equal(d.cpu().numpy(), o["terminated"], err_msg=ctx + " terminated")
        np.testing.assert_array_equal(eng.info["truncated"].cpu().numpy(), o["truncated"], err_msg=ctx + " truncated")
        np.testing.assert_array_equal(eng.info["out_of_bound"].cpu().numpy(), o["oob"], err_msg=ctx + " oob")
        np.testing.assert_array_equal(eng.info["step_count"].cpu().numpy(), o["step_count"], err_msg=ctx + " step_count")
        np.testing.assert_array_equal(eng.info["network_availability"].cpu().numpy().view(np.uint64),
                                      o["availability"].view(np.uint64), err_msg=ctx + " availability bits")
        if t % 25 == 24 or t == steps - 1:
            _compare_states(eng.get_state(), orc.get_state(), ctx)
    eng.close()


@pytest.mark.parametrize("trace", ["chain10_script", "toyctf_defender_s11"])
def test_general_layout_equals_packed_layout(trace, monkeypatch):
    """Small topologies run in the packed layout (16-bit sets, 4-byte rows); MCBS_NO_PACKED_SETS=1 (read at batch creation)
    forces the general layout every larger topology uses.  Same actions, same Philox draws: every output, the canonical state
    and the observations must be identical, which also ties the general layout's Chain-10 / ToyCtf behaviour to the traces."""
    from marlon_amd._abi import RNG_PHILOX
    _, sj = parity.load_trace(trace)
    topo = parity.topology_for(trace)
    E = 4096
    spec = parity.spec_from_json(sj, n_envs=E, auto_reset=True, rng_kind=RNG_PHILOX, seed=99, max_episode_steps=120)
    packed = _engine().BatchEngine(topo, spec)
    monkeypatch.setenv("MCBS_NO_PACKED_SETS", "1")
    general = _engine().BatchEngine(topo, spec)
    monkeypatch.delenv("MCBS_NO_PACKED_SETS")
    fields = list(parity.OBS_FIELDS)
    for t in range(200):
        a = packed.sample_actions(t % 4 != 0, seed=21, step=t)
        if t % 20 == 19:
            o1, o2 = packed.alloc_obs(fields), general.alloc_obs(fields)
            r1, d1 = packed.step_observe(a, o1)
            r2, d2 = general.step_observe(a, o2)
            for f in fields:
                assert packed.torch.equal(o1[f], o2[f]), f"step {t} obs {f}"
        else:
            r1, d1 = packed.step(a)
            r2, d2 = general.step(a)
        assert packed.torch.equal(r1, r2) and packed.torch.equal(d1, d2), f"step {t}"
        for k in ("raw_reward", "truncated", "out_of_bound", "step_count", "network_availability"):
            assert packed.torch.equal(packed.info[k], general.info[k]), f"step {t} info {k}"
        if t % 50 == 49:
            _compare_states(packed.get_state(), general.get_state(), f"{trace} step {t}")
    packed.close()
    general.close()


@pytest.mark.parametrize("trace", ["toyctf_defender_s11", "chain10_mix_s3", "chain4_defender_s21", "random24_defender_s51"])
def test_observations_match_oracle_batched(trace):
    """Observation kernels vs the oracle's observation for 512 envs in mixed states.  ToyCtf (connect rows of 70 bytes) takes
    the general mask kernels, Chain-10 / Chain-4 (96 / 48-byte rows) the masks fused into the per-env wavefront incl.
    mask_discrete, the 24-node topology has more (source, target) pairs than the fused path takes."""
    from marlon_amd._abi import RNG_PHILOX
    from oracle.oracle import Oracle
    _, sj = parity.load_trace(trace)
    topo = parity.topology_for(trace)
    E = 512
    spec = parity.spec_from_json(sj, n_envs=E, auto_reset=True, rng_kind=RNG_PHILOX, seed=5)
    eng = _engine().BatchEngine(topo, spec)
    orc = Oracle(topo, spec)
    fields = [f for f in parity.OBS_FIELDS]
    for t in range(60):
        a = eng.sample_actions(t % 3 != 0, seed=3, step=t)
        obs = eng.alloc_obs(fields + ["mask_discrete"])
        eng.step_observe(a, obs)
        oo = orc.alloc_obs(fields)
        orc.step(a.cpu().numpy(), obs=oo)
        for f in fields:
            np.testing.assert_array_equal(obs[f].cpu().numpy(), oo[f], err_msg=f"step {t} obs {f}")
        disc = np.concatenate([oo["mask_connect"].reshape(E, -1), oo["mask_local"].reshape(E, -1), oo["mask_remote"].reshape(E, -1)], axis=1)
        np.testing.assert_array_equal(obs["mask_discrete"].cpu().numpy(), disc, err_msg=f"step {t} mask_discrete")
    # rows padded to whole 128-byte lines (mcbs_set_mask_discrete_stride, what marlon_amd/wrappers.py allocates): the same mask in
    # [:, :A], nothing written beyond it
    A = eng.discrete_action_count()
    dense = eng.action_mask(eng.alloc_obs(["mask_discrete"]))["mask_discrete"].cpu().numpy()
    pad = (A + 127) // 128 * 128
    eng.set_mask_discrete_stride(pad)
    wide = eng.alloc_obs(["mask_discrete"])["mask_discrete"]
    assert tuple(wide.shape) == (E, pad)
    wide.fill_(7)
    eng.action_mask({"mask_discrete": wide})
    wn = wide.cpu().numpy()
    np.testing.assert_array_equal(wn[:, :A], dense, err_msg="padded mask_discrete rows")
    assert (wn[:, A:] == 7).all() and dense.any()
    eng.close()


@pytest.mark.parametrize("trace", ["toyctf_defender_s11", "chain10_mix_s3", "chain4_defender_s21"])
def test_quad_observation_kernel_equals_wavefront_per_env_and_oracle(trace, monkeypatch):
    """Observations of small topologies WITH mask fields have two writers: obs_quad_kernel (four envs per wavefront: small fields by
    16-lane groups, then the masks env after env; the default where mask rows are not whole cache lines, MCBS_QUAD_OBS=1 everywhere)
    and obs_small_kernel (a wavefront per env; MCBS_NO_QUAD_OBS=1).  Both against the oracle, every field of every step, on a batch
    that fills neither its last workgroup nor its last wavefront, with out-of-range actions (blank observations) in the mix, dense
    and line-padded mask_discrete rows, into buffers pre-filled with a sentinel (cyberbattle_env.py:643-677,753-773,859-933)."""
    from marlon_amd._abi import RNG_PHILOX
    from oracle.oracle import Oracle
    _, sj = parity.load_trace(trace)
    topo = parity.topology_for(trace)
    E = 515
    spec = parity.spec_from_json(sj, n_envs=E, auto_reset=True, rng_kind=RNG_PHILOX, seed=31, max_episode_steps=45)
    monkeypatch.setenv("MCBS_QUAD_OBS", "1")
    quad = _engine().BatchEngine(topo, spec)
    monkeypatch.delenv("MCBS_QUAD_OBS")
    monkeypatch.setenv("MCBS_NO_QUAD_OBS", "1")
    wave = _engine().BatchEngine(topo, spec)
    monkeypatch.delenv("MCBS_NO_QUAD_OBS")
    orc = Oracle(topo, spec)
    fields = [f for f in parity.OBS_FIELDS]
    A = quad.discrete_action_count()
    pad = (A + 127) // 128 * 128
    blanks = 0
    for t in range(100):
        if t == 50:                                            # second half: mask_discrete rows padded to whole 128-byte lines
            quad.set_mask_discrete_stride(pad)
            wave.set_mask_discrete_stride(pad)
        a = quad.sample_actions(t % 3 != 0, seed=13, step=t)
        if t % 7 == 2:
            a[::3, 1] = spec.maximum_node_count + 1            # beyond the discovered nodes: out of bound, the blank observation
        oq, ow = quad.alloc_obs(fields + ["mask_discrete"]), wave.alloc_obs(fields + ["mask_discrete"])
        for o in (oq, ow):
            for v in o.values():
                v.fill_(5)
        quad.step_observe(a, oq)
        wave.step_observe(a, ow)
        oo = orc.alloc_obs(fields)
        out = orc.step(a.cpu().numpy(), obs=oo)
        blanks += int(out["oob"].sum())
        for f in fields:
            np.testing.assert_array_equal(oq[f].cpu().numpy(), oo[f], err_msg=f"{trace} step {t} {f}: quad kernel vs oracle")
            np.testing.assert_array_equal(ow[f].cpu().numpy(), oo[f], err_msg=f"{trace} step {t} {f}: wavefront per env vs oracle")
        disc = np.concatenate([oo["mask_connect"].reshape(E, -1), oo["mask_local"].reshape(E, -1), oo["mask_remote"].reshape(E, -1)], axis=1)
        for name, o in (("quad", oq), ("wave", ow)):
            m = o["mask_discrete"].cpu().numpy()
            np.testing.assert_array_equal(m[:, :A], disc, err_msg=f"{trace} step {t} mask_discrete ({name})")
            assert (m[:, A:] == 5).all(), f"{trace} step {t}: {name} wrote into the padding"
    assert blanks > 50
    quad.close()
    wave.close()


@pytest.mark.parametrize("trace", ["toyctf_defender_s11", "chain4_defender_s21", "tiny_defender_s62", "sink_defender_s43"])
def test_observation_writers_over_observation_bounds(trace, monkeypatch):
    """The mask writers pick their shape from divisibility: connect rows of P*Cmax bytes that are / are not multiples of 16 or 4, per-source
    blocks of Nmax*P*Cmax bytes, masks whose length is / is not a whole number of 16-byte chunks, env bases at every 4-byte phase.  The
    same topology under a spread of (maximum_node_count, maximum_total_credentials) bounds — the observation's shape, not the
    simulation's — through both observation kernels (four envs per wavefront; a wavefront per env; beyond 16 nodes the region kernels),
    every field and the flat Discrete mask against the oracle (cyberbattle_env.py:643-677, action_masking.py:96-110)."""
    from marlon_amd._abi import RNG_PHILOX
    from oracle.oracle import Oracle
    _, sj = parity.load_trace(trace)
    topo = parity.topology_for(trace)
    n, c = int(topo.n_nodes), max(1, len(topo.triples))
    E = 67
    fields = [f for f in parity.OBS_FIELDS]
    tried = set()
    for nm, cm in ((n, c), (n + 1, c + 1), (n + 2, c + 3), (13, 7), (15, 9), (16, 16), (16, 5), (11, 11), (12, 6), (14, 10), (20, c + 2)):
        nm, cm = max(nm, n), max(cm, c)
        if (nm, cm) in tried:
            continue
        tried.add((nm, cm))
        spec = parity.spec_from_json(sj, n_envs=E, auto_reset=True, rng_kind=RNG_PHILOX, seed=77, max_episode_steps=30,
                                     maximum_node_count=nm, maximum_total_credentials=cm)
        monkeypatch.setenv("MCBS_QUAD_OBS", "1")
        quad = _engine().BatchEngine(topo, spec)
        monkeypatch.delenv("MCBS_QUAD_OBS")
        monkeypatch.setenv("MCBS_NO_QUAD_OBS", "1")
        wave = _engine().BatchEngine(topo, spec)
        monkeypatch.delenv("MCBS_NO_QUAD_OBS")
        orc = Oracle(topo, spec)
        A = quad.discrete_action_count()
        for t in range(24):
            a = quad.sample_actions(t % 4 != 3, seed=5, step=t)
            if t % 5 == 1:
                a[::4, 1] = nm + 2                             # out of bound: the blank observation
            oq, ow = quad.alloc_obs(fields + ["mask_discrete"]), wave.alloc_obs(fields + ["mask_discrete"])
            quad.step_observe(a, oq)
            wave.step_observe(a, ow)
            oo = orc.alloc_obs(fields)
            orc.step(a.cpu().numpy(), obs=oo)
            ctx = f"{trace} bounds ({nm}, {cm}) step {t}"
            disc = np.concatenate([oo["mask_connect"].reshape(E, -1), oo["mask_local"].reshape(E, -1), oo["mask_remote"].reshape(E, -1)], axis=1)
            assert disc.shape[1] == A
            for name, o in (("four envs per wavefront", oq), ("wavefront per env", ow)):
                for f in fields:
                    np.testing.assert_array_equal(o[f].cpu().numpy(), oo[f], err_msg=f"{ctx} {f} ({name})")
                np.testing.assert_array_equal(o["mask_discrete"].cpu().numpy(), disc, err_msg=f"{ctx} mask_discrete ({name})")
        quad.close()
        wave.close()
    assert len(tried) >= 8


def _many_ports_environment(n_ports):
    """Five nodes, `n_ports` ports: connect rows of n_ports * maximum_total_credentials bytes, the per-source block of the connect mask up
    to the LDS budget of the observation kernels and beyond it."""
    from marlon_amd import model as m
    allow = m.RulePermission.ALLOW
    ports = [f"P{i:02d}" for i in range(n_ports)]
    fw = lambda: m.FirewallConfiguration(incoming=[m.FirewallRule(p, allow) for p in ports], outgoing=[m.FirewallRule(p, allow) for p in ports])
    vuln = lambda kind, outcome: m.VulnerabilityInfo(description="", type=kind, outcome=outcome, cost=1.0)
    third = max(1, n_ports // 3)
    nodes = {
        "a": m.NodeInfo(services=[], value=0, agent_installed=True, properties=["X"], firewall=fw(), vulnerabilities={
            "find": vuln(m.VulnerabilityType.LOCAL, m.LeakedNodesId(["b", "c", "d"])),
            "keys": vuln(m.VulnerabilityType.LOCAL, m.LeakedCredentials([m.CachedCredential("b", ports[0], "k0"), m.CachedCredential("c", ports[third], "k1"),
                                                                         m.CachedCredential("d", ports[-1], "k2")]))}),
        "b": m.NodeInfo(services=[m.ListeningService(p, allowedCredentials=["k0"]) for p in ports[:third]], value=10, properties=["Y"], firewall=fw(),
                        vulnerabilities={"scan": vuln(m.VulnerabilityType.REMOTE, m.LeakedNodesId(["e"])),
                                         "more": vuln(m.VulnerabilityType.LOCAL, m.LeakedCredentials([m.CachedCredential("e", ports[1], "k3")]))}),
        "c": m.NodeInfo(services=[m.ListeningService(p, allowedCredentials=["k1"]) for p in ports[third:2 * third]], value=20, properties=["X", "Y"], firewall=fw()),
        "d": m.NodeInfo(services=[m.ListeningService(p, allowedCredentials=["k2"]) for p in ports[2 * third:]], value=30, firewall=fw()),
        "e": m.NodeInfo(services=[m.ListeningService(ports[1], allowedCredentials=["k3"])], value=50, properties=["Y"], firewall=fw()),
    }
    ids = m.Identifiers(properties=["X", "Y"], ports=ports, local_vulnerabilities=["find", "keys", "more"], remote_vulnerabilities=["scan"])
    return m.Environment(network=m.create_network(nodes), vulnerability_library={}, identifiers=ids)


@pytest.mark.parametrize("n_ports,cmax", [(15, 13), (15, 15), (15, 16), (13, 15), (9, 7)])
def test_observation_with_many_ports_at_the_lds_budget(n_ports, cmax, monkeypatch):
    """The per-source block of the connect mask (maximum_node_count * ports * maximum_total_credentials bytes, four variants in LDS per
    wavefront) is used up to what a workgroup's LDS holds next to the staging areas (mcbs_api.hip blk_cap): 15 ports x 13 credentials x
    16 nodes = 3 120 bytes fits (57 KB of LDS per workgroup), 15 x 15 x 16 = 3 600 does not and takes the period writer, 15 x 16 rows are
    whole 16-byte chunks.  Every field and the flat Discrete mask against the oracle, both observation kernels."""
    from marlon_amd import flatten as F
    from marlon_amd._abi import RNG_PHILOX, EnvSpec
    from oracle.oracle import Oracle
    topo = F.flatten(_many_ports_environment(n_ports))
    E = 131
    spec = EnvSpec(n_envs=E, maximum_node_count=16, maximum_total_credentials=cmax, maximum_discoverable_credentials_per_action=5,
                   attacker_goal=dict(own_atleast_percent=1.0), auto_reset=True, max_episode_steps=25, rng_kind=RNG_PHILOX, seed=3)
    monkeypatch.setenv("MCBS_QUAD_OBS", "1")
    quad = _engine().BatchEngine(topo, spec)
    monkeypatch.delenv("MCBS_QUAD_OBS")
    monkeypatch.setenv("MCBS_NO_QUAD_OBS", "1")
    wave = _engine().BatchEngine(topo, spec)
    monkeypatch.delenv("MCBS_NO_QUAD_OBS")
    orc = Oracle(topo, spec)
    fields = [f for f in parity.OBS_FIELDS]
    connects = 0
    for t in range(40):
        a = quad.sample_actions(t % 5 != 4, seed=9, step=t)
        oq, ow = quad.alloc_obs(fields + ["mask_discrete"]), wave.alloc_obs(fields + ["mask_discrete"])
        quad.step_observe(a, oq)
        wave.step_observe(a, ow)
        oo = orc.alloc_obs(fields)
        orc.step(a.cpu().numpy(), obs=oo)
        connects += int(oo["mask_connect"].sum())
        disc = np.concatenate([oo["mask_connect"].reshape(E, -1), oo["mask_local"].reshape(E, -1), oo["mask_remote"].reshape(E, -1)], axis=1)
        for name, o in (("four envs per wavefront", oq), ("wavefront per env", ow)):
            for f in fields:
                np.testing.assert_array_equal(o[f].cpu().numpy(), oo[f], err_msg=f"{n_ports} ports x {cmax} step {t} {f} ({name})")
            np.testing.assert_array_equal(o["mask_discrete"].cpu().numpy(), disc, err_msg=f"{n_ports} ports x {cmax} step {t} mask_discrete ({name})")
    assert connects > 1000
    quad.close()
    wave.close()


@pytest.mark.parametrize("n_nodes,seed", [(3, 1), (9, 2), (16, 3), (17, 4), (33, 5), (64, 6), (65, 7), (96, 11), (128, 8), (129, 9), (200, 10), (255, 12)])
def test_random_topologies_engine_vs_oracle(n_nodes, seed):
    """The config-5 generator at sizes on both sides of every layout boundary (packed / general at 16 nodes, 1 / 2 / 4 words
    per set at 64 / 128 nodes, list heads of 16 entries, LDS-staged vs global hot image), in-env defender with Philox draws,
    valid and uniform actions mixed: outputs every step, observations and the canonical state at intervals, against the oracle."""
    from marlon_amd import flatten as F, model
    from marlon_amd._abi import RNG_PHILOX, EnvSpec
    from marlon_amd.samples import random_net
    from oracle.oracle import Oracle
    topo = F.flatten(random_net.build(model, n_nodes, seed))
    E = 256 if n_nodes <= 64 else 96
    check_obs = n_nodes <= 33          # the connect mask is N*N*P*C bytes per env: 100+ MB each at 200 nodes
    spec = EnvSpec(n_envs=E, maximum_node_count=n_nodes, maximum_total_credentials=max(1, len(topo.triples)),
                   maximum_discoverable_credentials_per_action=8, attacker_goal=dict(own_atleast_percent=0.8),
                   defender=("scan_and_reimage", 0.4, 2, 3), maintain_sla=0.3, auto_reset=True, max_episode_steps=90,
                   rng_kind=RNG_PHILOX, seed=1234 + seed, env_id_base=77)
    eng = _engine().BatchEngine(topo, spec)
    orc = Oracle(topo, spec)
    fields = list(parity.OBS_FIELDS)
    for t in range(140):
        a = eng.sample_actions(t % 5 != 0, seed=5, step=t)
        an = a.cpu().numpy()
        ctx = f"random_net({n_nodes}, {seed}) step {t}"
        if t % 35 == 34 and check_obs:
            obs, oo = eng.alloc_obs(fields), orc.alloc_obs(fields)
            r, d = eng.step_observe(a, obs)
            o = orc.step(an, obs=oo)
            for f in fields:
                np.testing.assert_array_equal(obs[f].cpu().numpy(), oo[f], err_msg=f"{ctx} obs {f}")
        else:
            r, d = eng.step(a)
            o = orc.step(an)
        np.testing.assert_array_equal(r.double().cpu().numpy(), o["reward"], err_msg=ctx + " reward")
        np.testing.assert_array_equal(d.cpu().numpy(), o["terminated"], err_msg=ctx + " terminated")
        np.testing.assert_array_equal(eng.info["truncated"].cpu().numpy(), o["truncated"], err_msg=ctx + " truncated")
        np.testing.assert_array_equal(eng.info["network_availability"].cpu().numpy().view(np.uint64),
                                      o["availability"].view(np.uint64), err_msg=ctx + " availability bits")
        if t % 35 == 34 or t == 139:
            _compare_states(eng.get_state(), orc.get_state(), ctx)
    eng.close()


@pytest.mark.parametrize("trace,E", [("chain10_script", 4096), ("toyctf_defender_s11", 2048), ("chain100_defender_s31", 256),
                                     ("toyctf_randomevents_s81", 512)])
def test_step_many_equals_single_steps(trace, E):
    """mcbs_step_many (K steps in one launch) == K calls of mcbs_step: rewards, terminations, final state; episodes end, auto-reset
    and truncate inside the launch; packed and general layouts, in-env defenders with Philox draws."""
    from marlon_amd._abi import RNG_PHILOX
    _, sj = parity.load_trace(trace)
    topo = parity.topology_for(trace)
    spec = parity.spec_from_json(sj, n_envs=E, auto_reset=True, rng_kind=RNG_PHILOX, seed=31, max_episode_steps=60)
    one = _engine().BatchEngine(topo, spec)
    many = _engine().BatchEngine(topo, spec)
    K = 150
    t = one.torch
    ring = t.empty((K, E, 5), dtype=t.int32, device=one.device)
    r1 = t.empty((K, E), dtype=t.float32, device=one.device)
    d1 = t.empty((K, E), dtype=t.uint8, device=one.device)
    for k in range(K):
        one.sample_actions(k % 3 != 0, seed=9, step=k, out=ring[k])
        r, d = one.step(ring[k], with_info=False)
        r1[k], d1[k] = r, d
    r2, d2 = many.step_many(ring[:100])
    r3, d3 = many.step_many(ring[100:])
    assert t.equal(t.cat([r2, r3]), r1) and t.equal(t.cat([d2, d3]), d1)
    _compare_states(one.get_state(), many.get_state(), trace)
    one.close()
    many.close()


@pytest.mark.parametrize("trace,valid", [("chain10_script", True), ("toyctf_defender_s11", True), ("chain10_script", False)])
def test_rollout_random_equals_sample_then_step(trace, valid):
    """mcbs_rollout_random (sampling inside the step kernel) == mcbs_sample_actions + mcbs_step, step by step: same actions, rewards,
    terminations and final state."""
    from marlon_amd._abi import RNG_PHILOX
    _, sj = parity.load_trace(trace)
    topo = parity.topology_for(trace)
    E, K = 2048, 120
    spec = parity.spec_from_json(sj, n_envs=E, auto_reset=True, rng_kind=RNG_PHILOX, seed=77, max_episode_steps=50)
    one = _engine().BatchEngine(topo, spec)
    fused = _engine().BatchEngine(topo, spec)
    t = one.torch
    acts, rews, dones = [], [], []
    for k in range(K):
        a = one.sample_actions(valid, seed=123, step=1000 + k)
        r, d = one.step(a, with_info=False)
        acts.append(a.clone()); rews.append(r.clone()); dones.append(d.clone())
    r2, d2, a2 = fused.rollout_random(K, valid=valid, seed=123, first_step=1000, record_actions=True)
    assert t.equal(a2, t.stack(acts)) and t.equal(r2, t.stack(rews)) and t.equal(d2, t.stack(dones))
    _compare_states(one.get_state(), fused.get_state(), trace)
    r3, d3 = fused.step_many(t.stack(acts)[:5])        # the sampling mode does not leak into later scripted launches
    assert r3.shape == (5, E)
    one.close()
    fused.close()


def test_full_size_properties_chain10_65536():
    """BASELINE.json headline size (65 536 envs, Chain-10): determinism, shard invariance (two half batches with
    env_id_base = the full batch), and state invariants that hold for any action sequence."""
    from marlon_amd._abi import RNG_PHILOX
    _, sj = parity.load_trace("chain10_script")
    topo = parity.topology_for("chain10")
    E, steps = 65536, 120

    def run(n, base, ids):
        spec = parity.spec_from_json(sj, n_envs=n, auto_reset=True, rng_kind=RNG_PHILOX, seed=9, env_id_base=base, max_episode_steps=100)
        eng = _engine().BatchEngine(topo, spec)
        tot = eng.torch.zeros(n, dtype=eng.torch.float64, device=eng.device)
        dones = eng.torch.zeros(n, dtype=eng.torch.int64, device=eng.device)
        for t in range(steps):
            r, d = eng.step(eng.sample_actions(True, seed=1, step=t))
            tot += r.double()
            dones += d.long() + eng.info["truncated"].long()
        st = eng.get_state()
        eng.close()
        return tot.cpu().numpy(), dones.cpu().numpy(), st

    tot, dones, st = run(E, 0, None)
    tot2, dones2, st2 = run(E, 0, None)
    np.testing.assert_array_equal(tot, tot2)
    _compare_states(st, st2, "determinism")
    h0, d0, s0 = run(E // 2, 0, None)
    h1, d1, s1 = run(E // 2, E // 2, None)
    np.testing.assert_array_equal(np.concatenate([h0, h1]), tot)
    np.testing.assert_array_equal(np.concatenate([d0, d1]), dones)
    hdr, nodes, order, cache = st
    assert (dones > 0).all()                                         # every env was truncated at 100 steps at least once
    assert (hdr["n_discovered"] == nodes["discovered"].sum(axis=1)).all()
    assert (nodes["installed"] <= nodes["discovered"]).all()
    assert (nodes["privilege"][nodes["installed"] == 1] >= 1).all()
    assert (hdr["n_creds"] <= 11).all() and (hdr["step_count"] <= 100).all()
    assert tot.sum() > 0


def test_set_state_round_trip_and_continuation():
    """mcbs_set_state(mcbs_get_state()) is the identity, and a batch loaded from another batch's state continues
    exactly like it (ToyCtf + defender, nodes mid re-imaging included)."""
    from marlon_amd._abi import RNG_PHILOX
    _, sj = parity.load_trace("toyctf_marlon_s14")
    topo = parity.topology_for("toyctf")
    E = 1024
    spec = parity.spec_from_json(sj, n_envs=E, auto_reset=True, rng_kind=RNG_PHILOX, seed=21)
    a = _engine().BatchEngine(topo, spec)
    for t in range(90):
        a.step(a.sample_actions(True, seed=5, step=t))
    st = a.get_state()
    assert (st[1]["running"] == 0).any()                      # some nodes are being re-imaged at this point
    spec_b = parity.spec_from_json(sj, n_envs=E, auto_reset=True, rng_kind=RNG_PHILOX, seed=21)
    b = _engine().BatchEngine(topo, spec_b)
    b.set_state(*st)
    _compare_states(b.get_state(), st, "round trip")
    a.set_state(*st)
    _compare_states(a.get_state(), st, "identity")
    for t in range(90, 150):
        act = a.sample_actions(True, seed=5, step=t)
        ra, da = a.step(act)
        rb, db = b.step(act)
        np.testing.assert_array_equal(ra.cpu().numpy(), rb.cpu().numpy(), err_msg=f"step {t} reward")
        np.testing.assert_array_equal(da.cpu().numpy(), db.cpu().numpy(), err_msg=f"step {t} done")
        np.testing.assert_array_equal(a.info["network_availability"].cpu().numpy(), b.info["network_availability"].cpu().numpy())
    _compare_states(a.get_state(), b.get_state(), "continuation")
    a.close()
    b.close()


def test_random_events_availability_bits_with_non_dyadic_weights():
    """ExternalRandomEvents recomputes availability every step from the env's own service flags:
    sum over nodes of sla_weight * (1 + running service weights) / (1 + all service weights), every product and sum rounded
    separately like the reference's Python floats (actions.py:728-746).  Non-dyadic NODE and SERVICE weights and stopped services make
    a fused multiply-add visible in the last bit; engine vs oracle as uint64 bit patterns, Philox draws, 1 024 envs."""
    from marlon_amd import flatten as F, model
    from marlon_amd._abi import RNG_PHILOX, EnvSpec
    from marlon_amd.samples import kitchen_sink
    from oracle.oracle import Oracle
    env = kitchen_sink.build(model)
    nw = [0.1, 0.3, 0.7, 1.9, 2.3, 0.6, 1.1, 3.3]
    sw = [0.3, 0.6, 1.7, 0.9, 0.1, 2.2]
    k = 0
    for i, (_, info) in enumerate(env.nodes()):
        info.sla_weight = nw[i % len(nw)]
        for s in info.services:
            s.sla_weight = sw[k % len(sw)]
            k += 1
    assert k >= 4
    topo = F.flatten(env)
    assert int(topo.header()["avail_any_order"]) == 0
    E = 1024
    spec = EnvSpec(n_envs=E, maximum_node_count=topo.n_nodes, maximum_total_credentials=max(1, len(topo.triples)),
                   maximum_discoverable_credentials_per_action=8, attacker_goal=dict(own_atleast_percent=1.0), maintain_sla=0.0,
                   defender=("random_events",), auto_reset=True, max_episode_steps=120, rng_kind=RNG_PHILOX, seed=4242, env_id_base=9)
    eng = _engine().BatchEngine(topo, spec)
    orc = Oracle(topo, spec)
    distinct = set()
    for t in range(240):
        a = eng.sample_actions(t % 3 != 2, seed=8, step=t)
        r, d = eng.step(a)
        o = orc.step(a.cpu().numpy())
        av = eng.info["network_availability"].cpu().numpy()
        np.testing.assert_array_equal(av.view(np.uint64), o["availability"].view(np.uint64), err_msg=f"step {t} availability bits")
        np.testing.assert_array_equal(r.double().cpu().numpy(), o["reward"], err_msg=f"step {t} reward")
        np.testing.assert_array_equal(d.cpu().numpy(), o["terminated"], err_msg=f"step {t} terminated")
        distinct.update(np.unique(av).tolist())
    assert len(distinct) > 20 and min(distinct) < 0.9          # services were stopped: many different non-trivial sums were compared
    eng.close()


@pytest.mark.parametrize("trace", ["chain10_script", "toyctf_defender_s11", "random24_defender_s51"])
def test_lds_staged_hot_image_equals_default(trace, monkeypatch):
    """MCBS_LDS_TOPO=1 (read at batch creation) selects the step-kernel variant that stages the topology's hot image in LDS per
    workgroup — round 1's default, kept for the launch-shape comparison in profiles/round2_notes.md.  Same actions, same Philox draws:
    every output and the canonical state equal the default variant's (hot image through L1 / L2)."""
    from marlon_amd._abi import RNG_PHILOX
    _, sj = parity.load_trace(trace)
    topo = parity.topology_for(trace)
    E = 2048
    spec = parity.spec_from_json(sj, n_envs=E, auto_reset=True, rng_kind=RNG_PHILOX, seed=123, max_episode_steps=80)
    default = _engine().BatchEngine(topo, spec)
    monkeypatch.setenv("MCBS_LDS_TOPO", "1")
    staged = _engine().BatchEngine(topo, spec)
    monkeypatch.delenv("MCBS_LDS_TOPO")
    for t in range(160):
        a = default.sample_actions(t % 4 != 0, seed=17, step=t)
        r1, d1 = default.step(a)
        r2, d2 = staged.step(a)
        assert default.torch.equal(r1, r2) and default.torch.equal(d1, d2), f"step {t}"
        for k in ("raw_reward", "truncated", "out_of_bound", "step_count", "network_availability"):
            assert default.torch.equal(default.info[k], staged.info[k]), f"step {t} info {k}"
        if t % 40 == 39:
            _compare_states(default.get_state(), staged.get_state(), f"{trace} step {t}")
    default.close()
    staged.close()


@pytest.mark.parametrize("n_nodes,defender", [(100, True), (100, False), (256, True), (256, False)])
def test_cooperative_kernel_equals_one_lane_kernel_and_oracle(n_nodes, defender, monkeypatch):
    """Topologies with more than 64 nodes run mcbs_step on the G-lanes-per-env kernel (mcbs_step_coop.hip: G = 2 words per set at 100
    nodes, 4 at 256); MCBS_NO_COOP=1 keeps the one-lane-per-env kernel of mcbs_step.hip on the same state layout.  Both engines and
    the oracle are stepped side by side — every output of every step, the canonical state at intervals — on a batch that does not
    fill its last wavefront, with valid, uniform and out-of-range actions, truncation and auto-reset inside the launch, with and
    without the in-env defender (actions.py:325-423,524-606,700-746; defender.py:42-55)."""
    from marlon_amd import flatten as F, model
    from marlon_amd._abi import RNG_PHILOX, EnvSpec
    from marlon_amd.samples import random_net
    from oracle.oracle import Oracle
    topo = F.flatten(random_net.build(model, n_nodes, 21))
    E = 203
    kw = dict(defender=("scan_and_reimage", 0.5, 3, 2), maintain_sla=0.3) if defender else {}
    spec = EnvSpec(n_envs=E, maximum_node_count=n_nodes, maximum_total_credentials=max(1, len(topo.triples)),
                   maximum_discoverable_credentials_per_action=8, attacker_goal=dict(own_atleast_percent=0.6), auto_reset=True,
                   max_episode_steps=70, rng_kind=RNG_PHILOX, seed=99, env_id_base=1000, **kw)
    coop = _engine().BatchEngine(topo, spec)
    monkeypatch.setenv("MCBS_NO_COOP", "1")
    lane = _engine().BatchEngine(topo, spec)
    monkeypatch.delenv("MCBS_NO_COOP")
    orc = Oracle(topo, spec)
    ended = 0
    for t in range(180):
        a = coop.sample_actions(t % 6 != 5, seed=8, step=t)
        if t % 11 == 3:
            a[::5, 1] = n_nodes + 3                          # node index beyond the discovered nodes: the out-of-bound path
        an = a.cpu().numpy()
        r1, d1 = coop.step(a)
        r2, d2 = lane.step(a)
        o = orc.step(an)
        ctx = f"random_net({n_nodes}) defender={defender} step {t}"
        for name, x, y in (("reward", r1, r2), ("terminated", d1, d2)):
            np.testing.assert_array_equal(x.cpu().numpy(), y.cpu().numpy(), err_msg=f"{ctx} {name}: cooperative vs one-lane kernel")
        for k in coop.info:
            np.testing.assert_array_equal(coop.info[k].cpu().numpy().view(np.uint8), lane.info[k].cpu().numpy().view(np.uint8), err_msg=f"{ctx} info {k}")
        np.testing.assert_array_equal(r1.double().cpu().numpy(), o["reward"], err_msg=ctx + " reward vs oracle")
        np.testing.assert_array_equal(d1.cpu().numpy(), o["terminated"], err_msg=ctx + " terminated vs oracle")
        np.testing.assert_array_equal(coop.info["truncated"].cpu().numpy(), o["truncated"], err_msg=ctx + " truncated vs oracle")
        np.testing.assert_array_equal(coop.info["out_of_bound"].cpu().numpy(), o["oob"], err_msg=ctx + " oob vs oracle")
        np.testing.assert_array_equal(coop.info["network_availability"].cpu().numpy().view(np.uint64), o["availability"].view(np.uint64),
                                      err_msg=ctx + " availability bits vs oracle")
        ended += int(d1.sum()) + int(coop.info["truncated"].sum())
        if t % 30 == 29 or t == 179:
            _compare_states(coop.get_state(), lane.get_state(), ctx + " cooperative vs one-lane")
            _compare_states(coop.get_state(), orc.get_state(), ctx + " vs oracle")
    assert ended > E                                          # every env ended at least once, inside the launches
    coop.close()
    lane.close()
