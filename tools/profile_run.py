#!/usr/bin/env python3
"""One workload, a fixed number of launches, nothing else: the program rocprofv3 is pointed at (tools/profile_all.sh).

    python3 tools/profile_run.py step:<workload> [K]      K mcbs_step launches replayed from ONE hipGraph, twice (rehearsal + the replay bench.py
                                                          times: tools/workloads.py timed_leg) — the regime bench.py measures; recorded valid actions
    python3 tools/profile_run.py obs:<workload> [K]       K mcbs_observe launches of the whole observation (reference dtypes)
    python3 tools/profile_run.py discrete:<workload> [K]  K mcbs_observe launches of the small fields + mask_discrete (MaskablePPO path)
    python3 tools/profile_run.py logits:<workload> [K]    K mcbs_mask_logits launches (on-device mask -> logits, no mask materialised)
    python3 tools/profile_run.py wrapper:<workload> [K]   K AttackerVecEnv.step calls (Discrete actions, no mask materialised, hipGraph replay)
workload: headline | config2 | config3 | config4 | config5  (tools/workloads.py)

The recording rollout is ONE mcbs_rollout_random launch on a throw-away engine (the looping kernel, a different name), so every launch
of the workload's step kernel in the trace is a graph replay.  Prints one JSON line with the launch count so that the summariser can
cross-check."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tools import workloads as W  # noqa: E402

what, name = sys.argv[1].split(":")
K = int(sys.argv[2]) if len(sys.argv) > 2 else (200 if what == "step" else 10)
if what == "wrapper":
    import time
    from marlon_amd.wrappers import AttackerVecEnv
    topo_env, E, kw = W.wrapper_workload(name)
    venv = AttackerVecEnv(topo_env, E, discrete=True, materialize_masks=False, use_graph=os.environ.get("WRAPPER_GRAPH", "1") == "1", **kw)
    ref = AttackerVecEnv(topo_env, E, discrete=True, **kw)                     # supplies the masks a random valid policy samples from
    venv.reset(), ref.reset()
    g = torch.Generator(device=venv.engine.device)
    g.manual_seed(0)
    acts = []
    for _ in range(K + 5):
        m = ref.action_masks()
        a = torch.where(m, torch.rand(m.shape, generator=g, device=m.device), torch.full((1,), -1.0, device=m.device)).argmax(dim=1)
        ref.step(a)
        acts.append(a)
    ref.close()
    for a in acts[:5]:
        venv.step(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for a in acts[5:]:
        venv.step(a)
    e1.record()
    t1 = time.perf_counter()                                                  # everything is enqueued: the host's share
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    out = dict(what=what, workload=name, envs=E, launches=K, us_per_step=(t2 - t0) / K * 1e6, host_enqueue_us_per_step=(t1 - t0) / K * 1e6,
               device_us_per_step=e0.elapsed_time(e1) * 1e3 / K)
    if venv.use_graph:                      # the policy writes its actions into the wrapper's own buffer: no copy in front of the replay
        buf = venv.action_buffer
        buf.copy_(acts[-1])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(K):
            venv.step(buf)
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        out.update(in_place_us_per_step=(time.perf_counter() - t0) / K * 1e6, in_place_host_enqueue_us=(t1 - t0) / K * 1e6,
                   in_place_device_us=e0.elapsed_time(e1) * 1e3 / K)
    print(json.dumps(out))
    venv.close()
    raise SystemExit(0)
ring = W.record_ring(name, K if what == "step" else 40)
eng, topo, spec, desc = W.make_engine(name)
out = dict(what=what, workload=name, envs=eng.E, launches=K, desc=desc)
if what == "step":
    if os.environ.get("PROFILE_EAGER") == "1":          # the round-2 regime, for comparison: eager launches
        rewards = torch.empty((K, eng.E), dtype=torch.float32, device=eng.device)
        dones = torch.empty((K, eng.E), dtype=torch.uint8, device=eng.device)
        st = torch.cuda.current_stream().cuda_stream
        for t in range(K):
            assert eng.lib.mcbs_step(eng._h, ring[t].data_ptr(), rewards[t].data_ptr(), dones[t].data_ptr(), None, st) == 0
        torch.cuda.synchronize()
        out["regime"] = "eager"
    else:
        us, rewards, dones = W.graph_replay_us(eng, ring, K)
        out.update(regime="hipGraph replay (rehearsal + timed replay)", us_per_step_hip_events=us, step_kernel=W.STEP_KERNEL.get(name),
                   step_kernel_launches=2 * K)
    out["reward_sum"] = float(rewards.double().sum())
elif what in ("obs", "discrete"):
    fields = W.OBS_FIELDS if what == "obs" else W.OBS_FIELDS[:5] + ["mask_discrete"]
    us, bpe, obs = W.observe_us(eng, ring, fields, reps=K)
    out.update(us_per_observe=us, bytes_per_env=bpe)
elif what == "logits":
    for t in range(40):
        eng.step(ring[t], with_info=False)
    eng.observe(eng.alloc_obs(W.OBS_FIELDS[:5]))          # the digest the mask is rebuilt from is left by the last observation
    n_act = eng.discrete_action_count()
    logits = torch.zeros((eng.E, n_act), dtype=torch.float32, device=eng.device)
    for _ in range(K):
        eng.mask_logits(logits, fill=-1e8)
    torch.cuda.synchronize()
    out.update(bytes_per_env=4.0 * float((logits != 0).sum()) / eng.E, logits_bytes_per_env=n_act * 4)      # write-only: 4 B per masked-out action
else:
    raise SystemExit(f"unknown workload kind {what}")
print(json.dumps(out))
eng.close()
