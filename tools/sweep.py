#!/usr/bin/env python3
"""Batch-size sweep of the fused step kernel (Chain-10, attacker only): µs per launch vs number of envs.
Usage (GPU box): python tools/sweep.py [--steps 300]"""
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
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--sizes", type=str, default="1024,4096,16384,65536,262144,1048576")
ap.add_argument("--topology", type=str, default="chain10")
args = ap.parse_args()

if args.topology.startswith("chain"):
    n = int(args.topology[5:])
    topo = flatten.flatten(chainpattern.new_environment(n))
    kw = dict(maximum_node_count=n + 2, maximum_total_credentials=n + 2)
else:
    topo = flatten.flatten(toy_ctf.new_environment())
    kw = dict(maximum_node_count=12, maximum_total_credentials=10)
for E in [int(x) for x in args.sizes.split(",")]:
    spec = EnvSpec(n_envs=E, attacker_goal=dict(own_atleast_percent=1.0), auto_reset=True, max_episode_steps=2000, seed=1, **kw)
    eng = engine.BatchEngine(topo, spec)
    K = args.steps
    ring = torch.empty((K, E, 5), dtype=torch.int32, device=eng.device)
    for t in range(K):
        eng.sample_actions(True, seed=1, step=t, out=ring[t])
        eng.step(ring[t], with_info=False)
    torch.cuda.synchronize()
    eng.reset()
    eng.timing_enable(True)
    t0 = time.perf_counter()
    for t in range(K):
        eng.step(ring[t], with_info=False)
    ms, n = eng.timing_read()
    wall = time.perf_counter() - t0
    eng.timing_enable(False)
    print(json.dumps(dict(topology=args.topology, envs=E, kernel_us=ms * 1e3 / n, eager_wall_us=wall * 1e6 / K,
                          gsteps_per_s_kernel=E / (ms * 1e3 / n) / 1e3)))
    eng.close()
    del ring
    torch.cuda.empty_cache()
