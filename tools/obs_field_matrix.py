#!/usr/bin/env python3
"""Which fields of the observation cost what: us per mcbs_observe for subsets of the fields (tools/workloads.py observe_us), one JSON line
per subset.  python tools/obs_field_matrix.py [workload[@envs] ...]   (MCBS_NO_QUAD_OBS=1: a wavefront per env)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tools import workloads as W  # noqa: E402

SMALL = W.OBS_FIELDS[:5]
SUBSETS = {
    "small": SMALL, "connect": ["mask_connect"], "remote": ["mask_remote"], "local": ["mask_local"],
    "connect+remote": ["mask_remote", "mask_connect"], "masks": ["mask_local", "mask_remote", "mask_connect"],
    "small+connect": SMALL + ["mask_connect"], "all": W.OBS_FIELDS, "discrete": ["mask_discrete"], "small+discrete": SMALL + ["mask_discrete"],
}
for arg in sys.argv[1:] or ["config3", "headline"]:          # workload[@envs]
    name, _, n = arg.partition("@")
    ring = W.record_ring(name, 40, int(n or 0))
    eng, topo, spec, desc = W.make_engine(name, int(n or 0))
    first = True
    for label, fields in SUBSETS.items():
        us, bpe, obs = W.observe_us(eng, ring, fields, reps=20, advance=40 if first else 0)
        first = False
        print(json.dumps(dict(workload=name, envs=eng.E, fields=label, us=round(us, 2), bytes_per_env=bpe, TBps=round(bpe * eng.E / us / 1e6, 2),
                              quad="MCBS_NO_QUAD_OBS" not in os.environ)), flush=True)
        del obs
    eng.close()
    del ring
    torch.cuda.empty_cache()
