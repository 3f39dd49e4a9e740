#!/usr/bin/env python3
"""Observation tier of the headline workload and config 3 (ToyCtf), whole observation in the reference's dtypes: us per mcbs_observe,
bytes per env, fraction of the HBM peak (bench.py's `observe` leg without the rest).  Developer switches are read at batch creation:
    MCBS_NO_BLOCK_MASKS=1 python tools/bench_obs_configs.py      round 2's fused mask writers"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tools import workloads as W  # noqa: E402

for name in sys.argv[1:] or ["headline", "config3"]:
    ring = W.record_ring(name, 40)
    eng, topo, spec, desc = W.make_engine(name)
    for label, fields in (("all fields", W.OBS_FIELDS), ("small fields + mask_discrete", W.OBS_FIELDS[:5] + ["mask_discrete"])):
        us, bpe, obs = W.observe_us(eng, ring, fields, reps=20, advance=40 if label == "all fields" else 0)
        print(json.dumps(dict(workload=name, envs=eng.E, fields=label, us_per_observe=round(us, 2), bytes_per_env=bpe,
                              GBps=round(bpe * eng.E / us / 1e3, 1), frac=round(bpe * eng.E / us / 1e3 / W.HBM_PEAK_GBS, 3),
                              switches={k: v for k, v in os.environ.items() if k.startswith("MCBS_")})), flush=True)
        del obs
    eng.close()
    del ring
    torch.cuda.empty_cache()
