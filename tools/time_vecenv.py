#!/usr/bin/env python3
"""Host-side cost of the SB3 VecEnv surface (marlon_amd/vecenv.py MarlonVecEnv with numpy outputs, as Stable-Baselines3 consumes it):
wall time per `step` (device step + device-to-host copies of the observation + one info dict per env) and per
`np.stack(env_method("action_masks"))`, Chain-10, with and without materialised masks.  The device step itself is 40 us."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from marlon_amd.samples import chainpattern  # noqa: E402
from marlon_amd.vecenv import MarlonVecEnv  # noqa: E402
from marlon_amd.wrappers import AttackerVecEnv  # noqa: E402

for E in (256, 4096, 65536):
    for lean in (False, True):
        v = AttackerVecEnv(chainpattern.new_environment(10), E, maximum_node_count=12, maximum_total_credentials=12, discrete=True,
                           max_timesteps=50, materialize_masks=not lean)
        rows = []
        for numpy_outputs in (True, False):
            env = MarlonVecEnv(v, numpy_outputs=numpy_outputs)
            env.reset()
            acts = np.full(E, 12 * 12 * 7 * 12, dtype=np.int64)          # the first local action of node 0
            if not numpy_outputs:                                        # a trainer that keeps tensors on the device hands device actions over too
                acts = torch.as_tensor(acts, device=v.engine.device)
            for _ in range(3):
                env.step(acts)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            K = 10
            for _ in range(K):
                env.step(acts)                                           # infos NOT read: no per-env Python object is built
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / K
            t0 = time.perf_counter()
            for _ in range(K):
                infos = env.step(acts)[3]
                n_ep = sum(1 for info in infos if info.get("episode") is not None)   # BaseAlgorithm._update_info_buffer's loop
            torch.cuda.synchronize()
            dt_iter = (time.perf_counter() - t0) / K
            dm = 0.0
            if not lean and numpy_outputs:
                t0 = time.perf_counter()
                for _ in range(3):
                    np.stack(env.env_method("action_masks"))
                dm = (time.perf_counter() - t0) / 3
            rows.append(dict(envs=E, masks_materialised=not lean, numpy_outputs=numpy_outputs, step_ms=round(dt * 1e3, 3), step_ms_infos_iterated=round(dt_iter * 1e3, 3),
                             action_masks_ms=round(dm * 1e3, 2)))
        for r in rows:
            print(r, flush=True)
        v.close()
