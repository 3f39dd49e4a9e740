#!/usr/bin/env python3
"""Diagnostic: where does a wavefront of the step kernel spend its cycles?  Uses the -DMCBS_DIAG build
(marlon_amd/libmcbs_diag.so) which stamps s_memtime at the dependency-level boundaries.  Shares, not run times."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from marlon_amd import engine, flatten  # noqa: E402
from marlon_amd._abi import EnvSpec  # noqa: E402
from marlon_amd.samples import chainpattern  # noqa: E402

engine._lib = engine.load_library(os.path.join(os.path.dirname(engine.LIB_PATH), "libmcbs_diag.so"))
E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
topo = flatten.flatten(chainpattern.new_environment(10))
spec = EnvSpec(n_envs=E, maximum_node_count=12, maximum_total_credentials=12, attacker_goal=dict(own_atleast_percent=1.0),
               auto_reset=True, max_episode_steps=2000, seed=1)
eng = engine.BatchEngine(topo, spec)
lib = eng.lib
lib.mcbs_diag_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
waves = (E + 63) // 64
buf = torch.zeros((waves, 8), dtype=torch.int64, device=eng.device)
for t in range(300):
    a = eng.sample_actions(True, seed=1, step=t)
    if t == 299:
        lib.mcbs_diag_set_stamps(eng._h, buf.data_ptr())
    eng.step(a, with_info=False)
torch.cuda.synchronize()
s = buf.cpu().numpy().astype(np.int64)
names = ["entry->L1 issued", "L1 issued->landed(+LDS copy)", "->row landed", "->attacker logic+row store", "->all stores retired", "->reset/end"]
d = np.diff(s[:, :7], axis=1)
print("waves", waves, "median cycles per segment (s_memtime ticks):")
for i, n in enumerate(names):
    print(f"  {n:40s} median {np.median(d[:, i]):9.0f}  p90 {np.percentile(d[:, i], 90):9.0f}")
tot = s[:, 6] - s[:, 0]
print(f"  total per wave: median {np.median(tot):.0f} p90 {np.percentile(tot, 90):.0f}; first entry -> last end: {s[:, 6].max() - s[:, 0].min()} ticks")
print(f"  entry spread: {s[:, 0].max() - s[:, 0].min()} ticks;  realtime span {s[:,7].max()-s[:,7].min()} x10ns")
