#!/usr/bin/env python3
"""Step-only timing of BASELINE.json's configurations (per-GPU shard for the 8-GPU ones): hipGraph replay of recorded valid-action
batches on a fresh engine, HIP events on the launch stream (tools/workloads.py — the same code path bench.py's `configs` leg uses).
    python tools/bench_configs.py [K] [workload ...]        one JSON line per workload; developer switches (MCBS_LDS_TOPO,
    MCBS_STEP_BLOCK, MCBS_NO_PACKED_SETS ...) are read by the library at batch creation and echoed in the line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tools import workloads as W  # noqa: E402

args = sys.argv[1:]
K = int(args.pop(0)) if args and args[0].isdigit() else 300
names = args or ["config2", "headline", "config3", "config4", "config5"]
switches = {k: v for k, v in os.environ.items() if k.startswith("MCBS_")}
for name in names:
    n_envs = 0
    if "@" in name:                       # e.g. config4@65536: the workload at another batch size
        name, n = name.split("@")
        n_envs = int(n)
    ring = W.record_ring(name, K, n_envs=n_envs)
    eng, topo, spec, desc = W.make_engine(name, n_envs=n_envs)
    us, rewards, dones = W.graph_replay_us(eng, ring, K)
    print(json.dumps(dict(workload=name, envs=eng.E, nodes=topo.n_nodes, us_per_step=round(us, 3), G_env_steps_per_s=round(eng.E / us / 1e3, 3),
                          reward_sum=float(rewards.double().sum()), episodes_ended=int(dones.sum()), switches=switches, desc=desc)), flush=True)
    eng.close()
    del ring, rewards, dones
    torch.cuda.empty_cache()
