#!/usr/bin/env python3
"""Wrapper tier: one AttackerVecEnv.step (+ action_masks) per call — marlon's AttackerEnvWrapper / MaskedDiscreteAttackerWrapper
semantics for a whole batch — with a trivial random masked policy, Chain-10 and ToyCtf."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from marlon_amd.wrappers import AttackerVecEnv  # noqa: E402
from marlon_amd.samples import chainpattern, toy_ctf  # noqa: E402

for name, env, E, kw in (("chain10", chainpattern.new_environment(10), 65536, dict(maximum_node_count=12, maximum_total_credentials=12)),
                         ("toyctf", toy_ctf.new_environment(), 16384, dict(maximum_node_count=12, maximum_total_credentials=10))):
    for discrete, lean, graph in ((False, False, False), (True, False, False), (True, True, False), (True, False, True), (True, True, True)):
        # lean: no action mask is materialised; the policy's logits are masked in place by mcbs_mask_logits (timed separately below)
        venv = AttackerVecEnv(env, E, discrete=discrete, materialize_masks=not lean, use_graph=graph, **kw)
        ref = AttackerVecEnv(env, E, discrete=True, **kw) if lean else None      # supplies the masks the random policy samples from
        venv.reset()
        g = torch.Generator(device=venv.engine.device)
        g.manual_seed(0)

        def act():
            if discrete:
                m = (ref if lean else venv).action_masks()
                return torch.where(m, torch.rand(m.shape, generator=g, device=m.device), torch.full((1,), -1.0, device=m.device)).argmax(dim=1)
            nvec = torch.as_tensor(venv.nvec, device=venv.engine.device)
            return (torch.rand((E, len(nvec)), generator=g, device=nvec.device) * nvec).long()
        for _ in range(5):
            venv.step(act())
        torch.cuda.synchronize()
        K = 30
        acts = [act() for _ in range(K)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for a in acts:
            venv.step(a)
            if lean:
                ref.step(a)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        if lean:                               # time the lean wrapper alone on the same actions (the reference wrapper above only fed the policy)
            venv.reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for a in acts:
                venv.step(a)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / K
        t0 = time.perf_counter()
        for _ in range(K):
            venv.action_masks() if (discrete and not lean) else None
        torch.cuda.synchronize()
        dm = (time.perf_counter() - t0) / K
        row = dict(topology=name, envs=E, discrete=discrete, masks_materialised=not lean, graph=graph, step_us=dt * 1e6, action_masks_us=dm * 1e6,
                   M_env_steps_per_s=E / dt / 1e6)
        if lean:
            for dtype in (torch.float32, torch.bfloat16):
                logits = torch.zeros((E, venv.discrete_n), dtype=dtype, device=venv.engine.device)
                venv.mask_logits(logits)
                nbytes = float((logits != 0).sum()) * logits.element_size()        # write-only: the masked-out logits
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    venv.mask_logits(logits)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 100.0
                row[f"mask_logits_{str(dtype).split('.')[-1]}_us"] = us
                row[f"mask_logits_{str(dtype).split('.')[-1]}_GBps"] = nbytes / us / 1e3
                del logits
            ref.close()
        print(json.dumps(row), flush=True)
        venv.close()
