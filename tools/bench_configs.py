#!/usr/bin/env python3
"""Step-only timing of the BASELINE.json configurations that fit one GPU (per-GPU shard for the 8-GPU ones):
hipGraph replay of recorded valid-action batches, like bench.py.  Prints one JSON line per configuration."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from marlon_amd import engine, flatten, model  # noqa: E402
from marlon_amd._abi import EnvSpec  # noqa: E402
from marlon_amd.samples import chainpattern, random_net, toy_ctf  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 300
CONFIGS = [
    ("config2: Chain-10, 4096 envs, attacker only", chainpattern.new_environment(10), 4096,
     dict(maximum_node_count=12, maximum_total_credentials=12, attacker_goal=dict(own_atleast_percent=1.0))),
    ("headline: Chain-10, 65536 envs, attacker only", chainpattern.new_environment(10), 65536,
     dict(maximum_node_count=12, maximum_total_credentials=12, attacker_goal=dict(own_atleast_percent=1.0))),
    ("config3: ToyCtf, 16384 envs, ScanAndReimage(0.6,2,5), SLA 0.8", toy_ctf.new_environment(), 16384,
     dict(maximum_node_count=12, maximum_total_credentials=10, attacker_goal=dict(own_atleast=6, own_atleast_percent=1.0),
          maintain_sla=0.8, defender=("scan_and_reimage", 0.6, 2, 5))),
    ("config4 shard: Chain-100, 8192 envs (1/8 of 65536), ScanAndReimage", chainpattern.new_environment(100), 8192,
     dict(maximum_node_count=102, maximum_total_credentials=102, attacker_goal=dict(own_atleast_percent=1.0),
          defender=("scan_and_reimage", 0.6, 2, 5))),
    ("config4 whole on one GPU: Chain-100, 65536 envs, ScanAndReimage", chainpattern.new_environment(100), 65536,
     dict(maximum_node_count=102, maximum_total_credentials=102, attacker_goal=dict(own_atleast_percent=1.0),
          defender=("scan_and_reimage", 0.6, 2, 5))),
    ("config5 shard: Random-256, 16384 envs (1/8 of 131072), ScanAndReimage", random_net.build(model, 256, 0), 16384,
     dict(maximum_node_count=256, maximum_total_credentials=256, attacker_goal=dict(own_atleast_percent=1.0),
          maintain_sla=0.5, defender=("scan_and_reimage", 0.5, 4, 4))),
]
for name, env, E, kw in CONFIGS:
    topo = flatten.flatten(env)
    eng = engine.BatchEngine(topo, EnvSpec(n_envs=E, auto_reset=True, max_episode_steps=2000, seed=7, **kw))
    ring = torch.empty((K, E, 5), dtype=torch.int32, device=eng.device)
    for t in range(K):
        eng.sample_actions(True, seed=7, step=t, out=ring[t])
        eng.step(ring[t], with_info=False)
    torch.cuda.synchronize()
    eng.reset()
    rewards = torch.empty((K, E), dtype=torch.float32, device=eng.device)
    dones = torch.empty((K, E), dtype=torch.uint8, device=eng.device)
    lib, h = eng.lib, eng._h
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            s = torch.cuda.current_stream().cuda_stream
            for t in range(K):
                assert lib.mcbs_step(h, ring[t].data_ptr(), rewards[t].data_ptr(), dones[t].data_ptr(), None, s) == 0
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    hdr, _, _, _ = eng.get_state()
    state_bytes = int(lib.mcbs_state_record_bytes(h))
    print(json.dumps(dict(config=name, envs=E, us_per_step=dt / K * 1e6, G_env_steps_per_s=E * K / dt / 1e9,
                          reward_sum=float(rewards.sum()), episodes_ended=int(dones.sum()), nodes=topo.n_nodes)))
    eng.close()
    del ring, rewards, dones, g
    torch.cuda.empty_cache()
