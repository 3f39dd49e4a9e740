#!/usr/bin/env python3
"""Random-agent simulation on the device: sample + step per launch pair (mcbs_sample_actions, mcbs_step) against
mcbs_rollout_random (sampling inside the step kernel, 250 steps per launch), Chain-10 at 65 536 envs."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from marlon_amd import engine, flatten, simulate  # noqa: E402
from marlon_amd._abi import EnvSpec  # noqa: E402
from marlon_amd.samples import chainpattern  # noqa: E402

E, K = 65536, 1000
topo = flatten.flatten(chainpattern.new_environment(10))
spec = EnvSpec(n_envs=E, maximum_node_count=12, maximum_total_credentials=12, attacker_goal=dict(own_atleast_percent=1.0),
               auto_reset=True, max_episode_steps=2000, seed=3)
eng = engine.BatchEngine(topo, spec)
a = torch.empty((E, 5), dtype=torch.int32, device=eng.device)
for _ in range(20):
    eng.sample_actions(True, seed=1, step=0, out=a)
    eng.step(a, with_info=False)
eng.reset()
torch.cuda.synchronize()
t0 = time.perf_counter()
for t in range(K):
    eng.sample_actions(True, seed=1, step=t, out=a)
    eng.step(a, with_info=False)
torch.cuda.synchronize()
pair = (time.perf_counter() - t0) / K
eng.reset()
simulate.run_random_agents(eng, 50, seed=1)
eng.reset()
torch.cuda.synchronize()
t0 = time.perf_counter()
out = simulate.run_random_agents(eng, K, seed=1, chunk=250)
torch.cuda.synchronize()
fused = (time.perf_counter() - t0) / K
print(json.dumps(dict(envs=E, steps=K, sample_then_step_us=pair * 1e6, rollout_random_us=fused * 1e6,
                      G_env_steps_per_s_rollout=E / fused / 1e9, reward_sum=float(out["rewards"].sum()))))
