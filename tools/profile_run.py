#!/usr/bin/env python3
"""One workload, a fixed number of launches, nothing else: the program rocprofv3 is pointed at (tools/profile_all.sh).

    python3 tools/profile_run.py step:<workload> [K]      K mcbs_step launches (eager, recorded valid actions) on a fresh engine
    python3 tools/profile_run.py obs:<workload> [K]       K mcbs_observe launches of the whole observation (reference dtypes)
    python3 tools/profile_run.py discrete:<workload> [K]  K mcbs_observe launches of the small fields + mask_discrete (MaskablePPO path)
    python3 tools/profile_run.py logits:<workload> [K]    K mcbs_mask_logits launches (on-device mask -> logits, no mask materialised)
workload: headline | config2 | config3 | config4 | config5  (tools/workloads.py)

The recording rollout uses a throw-away engine; its launches are in the trace too (same kernels, same shapes), which only adds
samples to the per-kernel averages.  Prints one JSON line with the launch count so that the summariser can cross-check."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tools import workloads as W  # noqa: E402

what, name = sys.argv[1].split(":")
K = int(sys.argv[2]) if len(sys.argv) > 2 else (200 if what == "step" else 10)
ring = W.record_ring(name, K if what == "step" else 40)
eng, topo, spec, desc = W.make_engine(name)
out = dict(what=what, workload=name, envs=eng.E, launches=K, desc=desc)
if what == "step":
    rewards = torch.empty((K, eng.E), dtype=torch.float32, device=eng.device)
    dones = torch.empty((K, eng.E), dtype=torch.uint8, device=eng.device)
    st = torch.cuda.current_stream().cuda_stream
    for t in range(K):
        assert eng.lib.mcbs_step(eng._h, ring[t].data_ptr(), rewards[t].data_ptr(), dones[t].data_ptr(), None, st) == 0
    torch.cuda.synchronize()
    out["reward_sum"] = float(rewards.double().sum())
elif what in ("obs", "discrete"):
    fields = W.OBS_FIELDS if what == "obs" else W.OBS_FIELDS[:5] + ["mask_discrete"]
    us, bpe, obs = W.observe_us(eng, ring, fields, reps=K)
    out.update(us_per_observe=us, bytes_per_env=bpe)
elif what == "logits":
    for t in range(40):
        eng.step(ring[t], with_info=False)
    n_act = eng.discrete_action_count()
    logits = torch.zeros((eng.E, n_act), dtype=torch.float32, device=eng.device)
    for _ in range(K):
        eng.mask_logits(logits, fill=-1e8)
    torch.cuda.synchronize()
    out.update(bytes_per_env=n_act * 4 * 2)
else:
    raise SystemExit(f"unknown workload kind {what}")
print(json.dumps(out))
eng.close()
