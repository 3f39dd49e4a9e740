"""Shared helpers: load golden traces (captured from the imported reference by
oracle/refharness/gen_golden.py) and replay them through a stepper (the CPU oracle or the HIP engine)."""
from __future__ import annotations

import glob
import json
import os
import zlib

import numpy as np

from marlon_amd import flatten as F
from marlon_amd import model
from marlon_amd._abi import RNG_TAPE, EnvSpec
from marlon_amd.samples import (active_directory, chainpattern, generate_network, kitchen_sink, labelled_graph, random_net,
                                tinytoy, toy_ctf)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
OBS_FIELDS = ["scalars", "leaked_credentials", "credential_cache_matrix", "discovered_nodes_properties",
              "nodes_privilegelevel", "mask_local", "mask_remote", "mask_connect"]


def topology_for(trace_name: str) -> F.FlatTopology:
    """This build's own generators; tests/test_topology.py pins them to the reference's blobs."""
    if trace_name.startswith("chain100"):
        return F.flatten(chainpattern.new_environment(100))
    if trace_name.startswith("chain10"):
        return F.flatten(chainpattern.new_environment(10))
    if trace_name.startswith("chain4"):
        return F.flatten(chainpattern.new_environment(4))
    if trace_name.startswith("toyctf"):
        return F.flatten(toy_ctf.new_environment())
    if trace_name.startswith("sink_evict"):
        return F.flatten(kitchen_sink.build(model, entry_reimagable=True))
    if trace_name.startswith("sink"):
        return F.flatten(kitchen_sink.build(model))
    if trace_name.startswith("tinyad"):
        return F.flatten(active_directory.new_tiny_environment())
    if trace_name.startswith("tiny"):
        return F.flatten(tinytoy.new_environment())
    if trace_name[:2] == "ad" and trace_name[2].isdigit():
        return F.flatten(active_directory.new_random_environment(int(trace_name[2])))
    if trace_name.startswith("random_s"):
        return F.flatten(generate_network.new_environment(15, seed=int(trace_name.split("_")[1][1:])))
    if trace_name.startswith("labelled_s"):
        return F.flatten(labelled_graph.build(model, int(trace_name.split("_")[1][1:]), 6))
    if trace_name.startswith("random24"):
        return F.flatten(random_net.build(model, 24, 7))
    if trace_name.startswith("random100") or trace_name.startswith("random200"):
        return F.flatten(random_net.build(model, int(trace_name[6:9]), 7))
    raise KeyError(trace_name)


def trace_names():
    return sorted(n for n in (os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz"))) if not n.startswith("wrap_"))


def load_trace(name: str):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    spec = json.loads(bytes(z["spec_json"]).decode())
    return z, spec


def spec_from_json(spec: dict, n_envs: int = 1, auto_reset: bool = True, **over) -> EnvSpec:
    d = spec.get("defender")
    kw = dict(
        n_envs=n_envs, maximum_node_count=spec["maximum_node_count"],
        maximum_total_credentials=spec["maximum_total_credentials"],
        maximum_discoverable_credentials_per_action=spec["maximum_discoverable_credentials_per_action"],
        attacker_goal=spec["attacker_goal"], maintain_sla=spec["maintain_sla"],
        winning_reward=spec["winning_reward"], losing_reward=spec["losing_reward"],
        defender=None if d is None else tuple(d),
        auto_reset=auto_reset, rng_kind=RNG_TAPE)
    kw.update(over)
    return EnvSpec(**kw)


def replay(name: str, make_stepper, check_obs: bool = True, limit: int = 0):
    """Replay one golden trace on a single-env stepper and assert bit-exact agreement.

    make_stepper(topo, spec) must return an object with
        reset_observation(fields) -> dict of arrays [1, ...]
        step(actions[1,5], tape[1,dps] or None, want_obs) -> dict(reward, terminated, oob, step_count,
             availability, raw_reward, obs={field: [1, ...]}, order [1,N] u16, cache [1,C] u16)
    """
    z, sj = load_trace(name)
    topo = topology_for(name)
    spec = spec_from_json(sj)
    st = make_stepper(topo, spec)
    have_masks = "mask_connect" in z.files
    fields = [f for f in OBS_FIELDS if have_masks or not f.startswith("mask_")] if check_obs else []
    if check_obs:
        ro = st.reset_observation(OBS_FIELDS)
        for f in OBS_FIELDS:
            np.testing.assert_array_equal(ro[f][0], z["reset_" + f], err_msg=f"{name}: reset obs {f}")
    T = len(z["reward"]) if not limit else min(limit, len(z["reward"]))
    tape = z["tape"] if z["tape"].size else None
    for t in range(T):
        out = st.step(z["actions"][t:t + 1], None if tape is None else tape[t:t + 1], OBS_FIELDS if check_obs else None)
        ctx = f"{name} step {t} action {z['actions'][t].tolist()}"
        assert float(out["reward"][0]) == float(z["reward"][t]), f"{ctx}: reward {out['reward'][0]} != {z['reward'][t]}"
        assert float(out["raw_reward"][0]) == float(z["raw_reward"][t]), f"{ctx}: raw reward {out['raw_reward'][0]} != {z['raw_reward'][t]}"
        assert int(out["terminated"][0]) == int(z["terminated"][t]), f"{ctx}: terminated"
        assert int(out["step_count"][0]) == int(z["step_count"][t]), f"{ctx}: step_count"
        a = np.float64(out["availability"][0]).view(np.uint64)
        b = np.float64(z["availability"][t]).view(np.uint64)
        assert a == b, f"{ctx}: availability bits {out['availability'][0]!r} != {z['availability'][t]!r}"
        if check_obs:
            for f in fields:
                np.testing.assert_array_equal(out["obs"][f][0], z[f][t], err_msg=f"{ctx}: obs {f}")
            if not have_masks:
                crc = [zlib.crc32(np.ascontiguousarray(out["obs"][m][0]).tobytes()) for m in ("mask_local", "mask_remote", "mask_connect")]
                assert crc == z["mask_crc"][t].tolist(), f"{ctx}: mask crc"
        if "order" in out and not z["terminated"][t]:
            np.testing.assert_array_equal(out["order"][0], z["order"][t][:topo.n_nodes], err_msg=f"{ctx}: discovery order")
            np.testing.assert_array_equal(out["cache"][0], z["cache"][t], err_msg=f"{ctx}: credential cache order")
    return T
