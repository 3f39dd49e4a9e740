#!/usr/bin/env python3
"""Observation tier: time of mcbs_step_observe (attacker phase + observation kernels + defender phase) and of the
observation kernels alone, with bytes written per env (B_obs) against the HBM roofline."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from marlon_amd import engine, flatten  # noqa: E402
from marlon_amd._abi import EnvSpec  # noqa: E402
from marlon_amd.samples import chainpattern, toy_ctf  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--topology", default="chain10")
ap.add_argument("--fields", default="all")
args = ap.parse_args()
if args.topology == "toyctf":
    topo = flatten.flatten(toy_ctf.new_environment())
    kw = dict(maximum_node_count=12, maximum_total_credentials=10, attacker_goal=dict(own_atleast=6, own_atleast_percent=1.0),
              defender=("scan_and_reimage", 0.6, 2, 5), maintain_sla=0.8)
else:
    n = int(args.topology[5:])
    topo = flatten.flatten(chainpattern.new_environment(n))
    kw = dict(maximum_node_count=n + 2, maximum_total_credentials=n + 2, attacker_goal=dict(own_atleast_percent=1.0))
E, K = args.envs, args.steps
eng = engine.BatchEngine(topo, EnvSpec(n_envs=E, auto_reset=True, max_episode_steps=2000, seed=1, **kw))
fields = ["scalars", "leaked_credentials", "credential_cache_matrix", "discovered_nodes_properties", "nodes_privilegelevel",
          "mask_local", "mask_remote", "mask_connect"] if args.fields == "all" else args.fields.split(",")
obs = eng.alloc_obs(fields)
bytes_per_env = sum(v[0].numel() * v.element_size() for v in obs.values())
ring = torch.empty((K, E, 5), dtype=torch.int32, device=eng.device)
for t in range(K):
    eng.sample_actions(True, seed=1, step=t, out=ring[t])
    eng.step(ring[t], with_info=False)
eng.reset()
torch.cuda.synchronize()
for name, fn in (("step_observe", lambda t: eng.step_observe(ring[t], obs)), ("observe_only", lambda t: eng.observe(obs))):
    for t in range(5):
        fn(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(K):
        fn(t)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(json.dumps(dict(what=name, topology=args.topology, envs=E, us_per_call=dt * 1e6, obs_bytes_per_env=bytes_per_env,
                          GBps=bytes_per_env * E / dt / 1e9, frac_of_8TBps=bytes_per_env * E / dt / 8e12,
                          M_env_steps_per_s=E / dt / 1e6)))
