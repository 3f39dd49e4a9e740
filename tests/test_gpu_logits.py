"""mcbs_mask_logits (on-device action mask -> logits) against `where(oracle_mask, logits, fill)`:
the oracle's observation masks (env.py:643-677 restated in oracle/cbs_oracle.c) in MaskedDiscreteAttackerWrapper's order
(action_masking.py:96-110: connect | local | remote), 4 096 envs in mixed states incl. defender re-imaging between the
observation and the call (the mask is the OBSERVATION's, taken before the defender acts), float32 and bfloat16, aligned and
unaligned rows; and the wrapper without materialised masks."""
import numpy as np
import pytest

from tests import parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("trace", ["chain10_mix_s3", "toyctf_defender_s11", "random24_defender_s51"])
def test_mask_logits_equals_where_oracle_mask(trace):
    """fp32 and bf16 logits, dense rows, an offset view whose rows are neither 16-byte aligned nor a multiple of four long, and rows padded
    to whole 16-byte groups plus two groups (row_stride > actions: the padding is not the mask's to write)."""
    import torch
    from marlon_amd import engine
    from marlon_amd._abi import RNG_PHILOX
    from oracle.oracle import Oracle
    _, sj = parity.load_trace(trace)
    topo = parity.topology_for(trace)
    E = 4096 if topo.n_nodes <= 12 else 256            # the 24-node space has 24*24*P*C connect actions per env
    spec = parity.spec_from_json(sj, n_envs=E, auto_reset=True, rng_kind=RNG_PHILOX, seed=17, max_episode_steps=60)
    eng = engine.BatchEngine(topo, spec)
    orc = Oracle(topo, spec)
    A = eng.discrete_action_count()
    small = ["scalars", "leaked_credentials", "credential_cache_matrix", "discovered_nodes_properties", "nodes_privilegelevel"]
    obs = eng.alloc_obs(small)                          # NO mask field is requested from the observation (<= 16 nodes: the 16-lanes-per-env kernel)
    g = torch.Generator(device=eng.device).manual_seed(1)
    fill = -1e8
    for t in range(50):
        a = eng.sample_actions(t % 5 != 4, seed=9, step=t)
        check = t % 7 == 6 or t == 49
        oo = orc.alloc_obs(small + ["mask_local", "mask_remote", "mask_connect"]) if check else None
        if check:
            eng.step_observe(a, obs)
        else:
            eng.step(a)
        orc.step(a.cpu().numpy(), obs=oo)
        if not check:
            continue
        for f in small:
            np.testing.assert_array_equal(obs[f].cpu().numpy(), oo[f], err_msg=f"{trace} step {t} obs {f}")
        mask = np.concatenate([oo["mask_connect"].reshape(E, -1), oo["mask_local"].reshape(E, -1), oo["mask_remote"].reshape(E, -1)], axis=1) != 0
        assert mask.shape == (E, A)
        logits = torch.randn((E, A), generator=g, device=eng.device, dtype=torch.float32)
        ref = np.where(mask, logits.cpu().numpy(), np.float32(fill))
        out = eng.mask_logits(logits.clone(), fill)
        np.testing.assert_array_equal(out.cpu().numpy(), ref, err_msg=f"{trace} step {t} float32")
        lb = logits.to(torch.bfloat16)
        fb = torch.tensor(fill, dtype=torch.bfloat16)
        refb = torch.where(torch.as_tensor(mask), lb.cpu(), fb)
        outb = eng.mask_logits(lb.clone(), fill)
        assert torch.equal(outb.cpu().view(torch.int16), refb.view(torch.int16)), f"{trace} step {t} bfloat16"
        # rows that are neither 16-byte aligned nor a multiple of four long: a strided view into a wider buffer, offset by one element
        wide = torch.zeros((E, A + 3), device=eng.device, dtype=torch.float32)
        view = wide[:, 1:A + 1]
        view.copy_(logits)
        eng.mask_logits(view, fill)
        np.testing.assert_array_equal(view.cpu().numpy(), ref, err_msg=f"{trace} step {t} unaligned rows")
        assert float(wide[:, 0].abs().sum()) == 0.0 and float(wide[:, A + 1:].abs().sum()) == 0.0      # nothing outside the rows was touched
        # rows padded to a whole number of 16-byte groups (and two groups more): fp32 and bf16, padding untouched
        for dt, pad_to in ((torch.float32, 4), (torch.bfloat16, 8)):
            Ap = (A + pad_to - 1) // pad_to * pad_to + 2 * pad_to
            widep = torch.full((E, Ap), 7.0, device=eng.device, dtype=dt)
            vp = widep[:, :A]
            src = logits.to(dt)
            vp.copy_(src)
            eng.mask_logits(vp, fill)
            want = torch.where(torch.as_tensor(mask), src.cpu(), torch.tensor(fill, dtype=dt))
            assert torch.equal(vp.cpu().view(torch.int16 if dt == torch.bfloat16 else torch.int32),
                               want.view(torch.int16 if dt == torch.bfloat16 else torch.int32)), f"{trace} step {t} padded rows {dt}"
            assert bool((widep[:, A:] == 7.0).all()), f"{trace} step {t} padded rows {dt}: padding written"
    assert mask.any() and not mask.all()
    eng.close()


def test_wrapper_without_materialised_masks():
    """AttackerVecEnv(materialize_masks=False): same rewards / flags / small observation fields as the mask-writing wrapper, and
    mask_logits(logits) == where(action_masks() of that wrapper, logits, fill) at every step, auto-resets included."""
    import torch
    from marlon_amd import cyberbattle_env as ce
    from marlon_amd.samples import chainpattern
    from marlon_amd.wrappers import AttackerVecEnv
    E = 2048
    kw = dict(maximum_node_count=6, maximum_total_credentials=6, attacker_goal=ce.AttackerGoal(own_atleast_percent=1.0), max_timesteps=25, discrete=True)
    full = AttackerVecEnv(chainpattern.new_environment(4), E, **kw)
    lean = AttackerVecEnv(chainpattern.new_environment(4), E, materialize_masks=False, **kw)
    assert "connect" not in lean.observation and "connect" in full.observation
    with pytest.raises(RuntimeError, match="materialize_masks=False"):
        lean.action_masks()
    g = torch.Generator(device=full.engine.device).manual_seed(3)
    for t in range(70):
        m = full.action_masks()
        logits = torch.rand(m.shape, generator=g, device=m.device)
        masked = lean.mask_logits(logits.clone(), fill=-1.0)
        assert torch.equal(masked, torch.where(m, logits, torch.full_like(logits, -1.0))), f"step {t}"
        actions = masked.argmax(dim=1)                   # a masked-greedy policy on random scores = uniform over the valid actions
        if t % 9 == 4:
            actions[::7] = full.discrete_n - 1           # undiscovered indices: intercepted, the env keeps its last observation (and its digest)
        o1, r1, te1, tr1, i1 = full.step(actions)
        o2, r2, te2, tr2, i2 = lean.step(actions)
        assert torch.equal(r1, r2) and torch.equal(te1, te2) and torch.equal(tr1, tr2) and torch.equal(i1["invalid_action"], i2["invalid_action"])
        for k in o2:
            assert torch.equal(o1[k], o2[k]), f"step {t} obs {k}"
    full.close(); lean.close()


def test_mask_logits_refuses_stale_digests():
    """mcbs_mask_logits rebuilds the mask from the digest the last observation left per env (the local block through the live discovery
    list): it must refuse (MCBS_ESTATE) while no observation has been taken — after creation, a whole-batch reset, mcbs_set_state — and
    while envs reset by mask have not been re-observed; it works again after the matching observation."""
    import torch
    from marlon_amd import engine
    from marlon_amd._abi import EnvSpec
    from marlon_amd.flatten import flatten
    from marlon_amd.samples import chainpattern
    topo = flatten(chainpattern.new_environment(4))
    eng = engine.BatchEngine(topo, EnvSpec(n_envs=64, maximum_node_count=6, maximum_total_credentials=6, attacker_goal=dict(own_atleast_percent=1.0)))
    logits = torch.zeros((64, eng.discrete_action_count()), device=eng.device)
    small = eng.alloc_obs(["scalars", "nodes_privilegelevel"])
    with pytest.raises(engine.McbsError, match="no observation"):
        eng.mask_logits(logits)
    eng.observe(small)
    eng.mask_logits(logits)
    for t in range(5):
        eng.step(eng.sample_actions(True, seed=1, step=t))
    eng.mask_logits(logits)                                  # steps do not invalidate: the mask is the LAST OBSERVATION's by definition
    mask = torch.zeros(64, dtype=torch.uint8, device=eng.device)
    mask[::3] = 1
    eng.reset(mask)
    with pytest.raises(engine.McbsError, match="reset by mask"):
        eng.mask_logits(logits)
    eng.observe(small, env_mask=mask)
    eng.mask_logits(logits)
    hdr, nodes, order, cache = eng.get_state()
    eng.set_state(hdr, nodes, order, cache)
    with pytest.raises(engine.McbsError, match="no observation"):
        eng.mask_logits(logits)
    eng.observe(small)
    eng.mask_logits(logits)
    eng.reset()
    with pytest.raises(engine.McbsError, match="no observation"):
        eng.mask_logits(logits)
    eng.close()
