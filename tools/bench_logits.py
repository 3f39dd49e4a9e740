#!/usr/bin/env python3
"""mcbs_mask_logits alone: us per launch on the headline batch at a few points of an episode (the share of allowed actions grows
with the attacker's progress), fp32 and bf16 logits.  `MCBS_LOGITS_GRID` (developer switch) = workgroups walking the envs."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tools import workloads as W  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "headline"
ring = W.record_ring(name, 200)
eng, topo, spec, desc = W.make_engine(name)
obs = eng.alloc_obs(W.OBS_FIELDS[:5])
A = eng.discrete_action_count()
t = 0
for upto in (1, 40, 200):
    while t < upto:
        eng.step(ring[t], with_info=False)
        t += 1
    eng.observe(obs)
    row = dict(workload=name, envs=eng.E, actions=A, after_steps=upto, grid=os.environ.get("MCBS_LOGITS_GRID", "default"))
    for dtype in (torch.float32, torch.bfloat16):
        logits = torch.zeros((eng.E, A), dtype=dtype, device=eng.device)
        eng.mask_logits(logits, fill=-1e8)
        allowed = float((logits == 0).sum()) / eng.E
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            eng.mask_logits(logits, fill=-1e8)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100.0
        k = str(dtype).split(".")[-1]
        row.update({"allowed_per_env": allowed, f"{k}_us": round(us, 1), f"{k}_written_GBps": round((A - allowed) * logits.element_size() * eng.E / us / 1e3, 1)})
        if upto == 1:                                   # reference points on the same buffer: a plain fill, and torch's own where()
            for nm, fn in (("fill", lambda: logits.fill_(-1e8)), ("torch_where", lambda: torch.where(mask, logits, neg, out=logits))):
                if nm == "torch_where":
                    mask, neg = (logits == 0), torch.full((1,), -1e8, dtype=dtype, device=eng.device)
                fn()
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                row[f"{k}_{nm}_us"] = round(e0.elapsed_time(e1) * 100.0, 1)
            del mask, neg
        del logits
    print(json.dumps(row), flush=True)
eng.close()
