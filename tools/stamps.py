#!/usr/bin/env python3
"""Diagnostic: where does a wavefront of the step kernel spend its cycles?  Uses the -DMCBS_DIAG build
(marlon_amd/libmcbs_diag.so, `make -C marlon_amd/csrc diag`) which stamps s_memtime at the dependency-level boundaries.
Shares, not run times: the stamps drain the memory queue.     python tools/stamps.py [workload ...]   (tools/workloads.py names)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from marlon_amd import engine  # noqa: E402
from tools import workloads as W  # noqa: E402

engine._lib = engine.load_library(os.path.join(os.path.dirname(engine.LIB_PATH), "libmcbs_diag.so"))
names = ["entry->L1 issued", "L1 issued->landed", "->row + tables landed", "->attacker logic+row store", "->all stores retired", "->reset/end"]
for wl in (sys.argv[1:] or ["headline"]):
    eng, topo, spec, desc = W.make_engine(wl)
    lib = eng.lib
    lib.mcbs_diag_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
    waves = (eng.E + 63) // 64
    buf = torch.zeros((waves, 8), dtype=torch.int64, device=eng.device)
    T = 120
    for t in range(T):
        a = eng.sample_actions(True, seed=1, step=t)
        if t == T - 1:
            lib.mcbs_diag_set_stamps(eng._h, buf.data_ptr())
        eng.step(a, with_info=False)
    torch.cuda.synchronize()
    s = buf.cpu().numpy().astype(np.int64)
    d = np.diff(s[:, :7], axis=1)
    print(f"{wl}: {desc}, {eng.E} envs, {waves} waves; median s_memtime ticks per segment:")
    for i, n in enumerate(names):
        print(f"  {n:32s} median {np.median(d[:, i]):9.0f}  p90 {np.percentile(d[:, i], 90):9.0f}")
    tot = s[:, 6] - s[:, 0]
    print(f"  total per wave: median {np.median(tot):.0f} p90 {np.percentile(tot, 90):.0f}; first entry -> last end: {s[:, 6].max() - s[:, 0].min()} ticks; "
          f"entry spread {s[:, 0].max() - s[:, 0].min()} ticks; realtime span {s[:, 7].max() - s[:, 7].min()} x10ns")
    eng.close()
