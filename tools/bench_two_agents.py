#!/usr/bin/env python3
"""One joint turn of marlon's two learned agents for a batch — AttackerVecEnv.step then DefenderVecEnv.step on the shared environment
batch (marl_algorithm.run_episode's order, marl_algorithm.py:197-250) — us per turn, ToyCtf, random valid-ish policies on the device.
    python tools/bench_two_agents.py [envs]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from marlon_amd.samples import toy_ctf  # noqa: E402
from marlon_amd.wrappers import AttackerVecEnv, DefenderVecEnv  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
for graph in (False, True):
    att = AttackerVecEnv(toy_ctf.new_environment(), E, maximum_node_count=12, maximum_total_credentials=10, discrete=True, learned_defender=True,
                         materialize_masks=False, max_timesteps=200)
    dfd = DefenderVecEnv(att, max_timesteps=200, use_graph=graph)
    dev = att.engine.device
    g = torch.Generator(device=dev).manual_seed(0)
    a_acts = [torch.randint(0, att.discrete_n, (E,), generator=g, device=dev) for _ in range(8)]
    nv = torch.as_tensor(dfd.nvec, device=dev)
    d_acts = [(torch.rand((E, 12), generator=g, device=dev) * nv).long() for _ in range(8)]
    for i in range(8):
        att.step(a_acts[i]); dfd.step(d_acts[i])
    torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter()
    for i in range(K):
        att.step(a_acts[i & 7])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(K):
        dfd.step(d_acts[i & 7])
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for i in range(K):
        att.step(a_acts[i & 7]); dfd.step(d_acts[i & 7])
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(json.dumps(dict(envs=E, defender_graph=graph, attacker_step_us=round((t1 - t0) / K * 1e6, 1), defender_step_us=round((t2 - t1) / K * 1e6, 1),
                          joint_turn_us=round((t3 - t2) / K * 1e6, 1), attacker_launches=att.engine.wrapper_step_launches(False))), flush=True)
    att.close()
