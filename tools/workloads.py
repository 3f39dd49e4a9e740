"""BASELINE.json's workloads as the benchmark and the profiling tools run them (shared by bench.py, tools/profile_run.py and
tools/bench_configs.py so that a rocprofv3 summary under profiles/ and a bench.py line describe the same launches).

A workload = (topology, per-GPU env count, EnvSpec keywords).  Names:
    headline  Chain-10, 65 536 envs, attacker only                                   (BASELINE.json metric, configs[1] shape)
    config2   Chain-10, 4 096 envs, attacker only
    config3   ToyCtf, 16 384 envs, ScanAndReimage(0.6, 2, 5), SLA 0.80, own_atleast 6
    config4   Chain-100 (N 102, C 102), 8 192 envs = one GPU's shard of 65 536, ScanAndReimage(0.6, 2, 5)
    config5   Random-256 (this build's config-5 generator, seed 0), 16 384 envs = one GPU's shard of 131 072, ScanAndReimage(0.5, 4, 4)
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
OBS_FIELDS = ["scalars", "leaked_credentials", "credential_cache_matrix", "discovered_nodes_properties", "nodes_privilegelevel",
              "mask_local", "mask_remote", "mask_connect"]


# the fused-step kernel each workload's mcbs_step launches (rocprofv3's name without the argument list): profiles/ and bench.py agree on it
STEP_KERNEL = {
    "headline": "mcbs::step_kernel<0, 0, false, 0>",     # <whole step, packed sets, hot image through L1 / L2, no defender>
    "config2": "mcbs::step_kernel<0, 0, false, 0>",
    "config3": "mcbs::step_kernel<0, 0, false, 1>",      # packed, ScanAndReimage
    "config4": "mcbs::step_coop_kernel<2, 1>",           # two words per set: two lanes per env (mcbs_step_coop.hip)
    "config5": "mcbs::step_coop_kernel<4, 1>",           # four words per set: four lanes per env
}


def workload(name: str):
    from marlon_amd import flatten, model
    from marlon_amd.samples import chainpattern, random_net, toy_ctf
    if name in ("headline", "config2"):
        return (flatten.flatten(chainpattern.new_environment(10)), 65536 if name == "headline" else 4096,
                dict(maximum_node_count=12, maximum_total_credentials=12, attacker_goal=dict(own_atleast_percent=1.0)),
                "CyberBattleChain size=10, attacker only")
    if name == "config3":
        return (flatten.flatten(toy_ctf.new_environment()), 16384,
                dict(maximum_node_count=12, maximum_total_credentials=10, attacker_goal=dict(own_atleast=6, own_atleast_percent=1.0),
                     maintain_sla=0.8, defender=("scan_and_reimage", 0.6, 2, 5)),
                "CyberBattleToyCtf, attacker + ScanAndReimage(0.6, 2, 5), SLA 0.80")
    if name == "config4":
        return (flatten.flatten(chainpattern.new_environment(100)), 8192,
                dict(maximum_node_count=102, maximum_total_credentials=102, attacker_goal=dict(own_atleast_percent=1.0),
                     defender=("scan_and_reimage", 0.6, 2, 5)),
                "CyberBattleChain size=100, attacker + ScanAndReimage(0.6, 2, 5); one GPU's shard (1/8) of 65 536 envs")
    if name == "config5":
        topo = flatten.flatten(random_net.build(model, 256, 0))
        return (topo, 16384,
                dict(maximum_node_count=256, maximum_total_credentials=256, maximum_discoverable_credentials_per_action=8,
                     attacker_goal=dict(own_atleast_percent=1.0), maintain_sla=0.5, defender=("scan_and_reimage", 0.5, 4, 4)),
                "Random 256-node topology, attacker + ScanAndReimage(0.5, 4, 4); one GPU's shard (1/8) of 131 072 envs")
    raise KeyError(name)


def wrapper_workload(name: str):
    """(reference-style environment, envs, AttackerVecEnv keyword arguments) of the wrapper tier's two workloads."""
    from marlon_amd.samples import chainpattern, toy_ctf
    if name == "headline":
        return chainpattern.new_environment(10), 65536, dict(maximum_node_count=12, maximum_total_credentials=12)
    if name == "config3":
        return toy_ctf.new_environment(), 16384, dict(maximum_node_count=12, maximum_total_credentials=10)
    raise KeyError(name)


def make_engine(name: str, n_envs: int = 0, env_id_base: int = 0, device: str = "cuda:0", max_episode_steps: int = 2000, seed: int = 7):
    from marlon_amd import engine
    from marlon_amd._abi import EnvSpec
    topo, E, kw, desc = workload(name)
    E = n_envs or E
    spec = EnvSpec(n_envs=E, auto_reset=True, max_episode_steps=max_episode_steps, seed=seed, env_id_base=env_id_base, **kw)
    return engine.BatchEngine(topo, spec, device=device), topo, spec, desc


def record_ring(name: str, K: int, n_envs: int = 0, env_id_base: int = 0, device: str = "cuda:0", seed: int = 7):
    """K batches of valid random actions (the distribution of sample_valid_action) recorded by an untimed rollout of a
    THROW-AWAY engine of the same workload: a fresh (or rewound) engine with the same seed then replays exactly the recorded trajectory,
    defender draws included (Philox is keyed by (seed, global env id, episode, step)).  ONE launch of the looping step kernel with
    the random agent sampled inside it (mcbs_rollout_random: the actions mcbs_sample_actions(valid, seed, t) + mcbs_step would give) —
    a different kernel from the one mcbs_step launches, so a rocprofv3 trace of the caller holds no recording launches of that one."""
    import torch
    eng, _, _, _ = make_engine(name, n_envs, env_id_base, device, seed=seed)
    ring = eng.rollout_random(K, valid=True, seed=seed, first_step=0, record_actions=True)[2]
    torch.cuda.synchronize()
    eng.close()
    return ring


def graph_replay_us(eng, ring, K: int):
    """us per mcbs_step launch: the K recorded steps captured into one hipGraph, rehearsed once, the engine rewound, and replayed with
    HIP events on the launch stream around the replay (timed_leg).  Returns (us_per_step, rewards[K, E], dones[K, E])."""
    leg = timed_leg(eng, ring, 0, K, graph=True, barrier=lambda: None, restore=eng.rewind)
    return leg["region_us"], leg["rewards"], leg["dones"]


def timed_leg(eng, ring, Wm: int, K: int, graph: bool, barrier, restore):
    """W warm-up steps (ring[:W]) then EXACTLY K timed steps (ring[W:W+K]) of mcbs_step on `eng`, from the state restore() leaves.
    Both step sequences are captured as hipGraphs (graph=True) and REHEARSED once, untimed; restore() then puts the engine back to the
    start state, the warm-up graph is replayed, and the K timed steps run between barrier + synchronize on both sides with HIP events
    on the launch stream around them.  Returns elapsed_s (host clock, this rank), region_us (event time / K), rewards / dones [K, E]
    of the timed replay, rehearsal_equal (the rehearsal's per-env sums equal the timed replay's: the replay is deterministic)."""
    import torch
    E, dev = eng.E, eng.device
    lib, h = eng.lib, eng._h
    rewards = torch.empty((K, E), dtype=torch.float32, device=dev)
    dones = torch.empty((K, E), dtype=torch.uint8, device=dev)
    wr = torch.empty((max(Wm, 1), E), dtype=torch.float32, device=dev)
    wd = torch.empty((max(Wm, 1), E), dtype=torch.uint8, device=dev)

    def run_steps(t0: int, n: int, r, d, stream: int) -> None:
        for t in range(n):
            if lib.mcbs_step(h, ring[t0 + t].data_ptr(), r[t].data_ptr(), d[t].data_ptr(), None, stream) != 0:
                raise RuntimeError(lib.mcbs_last_error().decode())

    def capture(t0: int, n: int, r, d):
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                run_steps(t0, n, r, d, torch.cuda.current_stream().cuda_stream)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        return g          # capture does not execute: the engine's state is untouched

    if graph:
        g_warm = capture(0, Wm, wr, wd) if Wm else None
        g_timed = capture(Wm, K, rewards, dones)
        warm = (lambda: g_warm.replay()) if Wm else (lambda: None)
        timed = lambda: g_timed.replay()
    else:
        st = torch.cuda.current_stream().cuda_stream
        warm = lambda: run_steps(0, Wm, wr, wd, st)
        timed = lambda: run_steps(Wm, K, rewards, dones, st)

    # rehearsal (untimed) from the start state, then back to it
    restore()
    warm()
    timed()
    torch.cuda.synchronize()
    reh_r, reh_d = rewards.double().sum(dim=0), dones.long().sum(dim=0)
    rewards.zero_()
    dones.zero_()
    restore()
    # W untimed warm-up steps
    warm()
    torch.cuda.synchronize()
    # timed region: exactly K steps.  (The two HIP events are created and recorded once beforehand: torch creates the event object on its
    # first record(), which cost 13 + 7 us of host time inside a region of 100 us of device work — tools/short_run_env.sh)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    ev1.record()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()         # HIP events on the launch stream (torch's current stream IS the stream the K launches go to)
    t1 = time.perf_counter()
    timed()
    t2 = time.perf_counter()
    ev1.record()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    elapsed = t4 - t0                                 # this rank's K steps, device work drained; the MAX over ranks is the job's time
    barrier()                                         # (the closing barrier brackets the region; its own latency is not part of the K steps)
    region_us = ev0.elapsed_time(ev1) * 1e3 / K       # device time per launch over the timed region, launch gaps included
    same = bool(torch.equal(rewards.double().sum(dim=0), reh_r) and torch.equal(dones.long().sum(dim=0), reh_d))
    host = {"event_record_us": (t1 - t0) * 1e6, "enqueue_us": (t2 - t1) * 1e6, "event_record2_us": (t3 - t2) * 1e6, "synchronize_us": (t4 - t3) * 1e6,
            "device_region_us": region_us * K, "host_region_us": elapsed * 1e6}
    return {"elapsed_s": elapsed, "region_us": region_us, "rewards": rewards, "dones": dones, "rehearsal_equal": same, "host_breakdown": host}


def observe_us(eng, ring, fields, reps: int = 20, advance: int = 40):
    """us per mcbs_observe of `fields` (int8 masks and int32 fields in the reference's layout) on a batch that has been advanced by
    `advance` recorded steps, HIP events on the launch stream.  Returns (us, bytes per env, obs dict)."""
    import torch
    if "mask_discrete" in fields and not eng.mask_discrete_stride:
        # the flat Discrete mask as marlon_amd/wrappers.py allocates it: rows padded to whole 128-byte lines (mcbs_set_mask_discrete_stride)
        eng.set_mask_discrete_stride((eng.discrete_action_count() + 127) // 128 * 128)
    obs = eng.alloc_obs(fields)
    bytes_per_env = sum((eng.discrete_action_count() if k == "mask_discrete" else v[0].numel()) * v.element_size() for k, v in obs.items())
    for t in range(min(advance, ring.shape[0])):
        eng.step(ring[t], with_info=False)
    for _ in range(3):
        eng.observe(obs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        eng.observe(obs)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps, bytes_per_env, obs


def csrc_sha256() -> str:
    """Hash of every source the native library is built from: ties a counter file under profiles/ to the kernels it was taken on
    (the GPU box has no .git, so a commit id is not available there)."""
    h = hashlib.sha256()
    d = os.path.join(REPO, "marlon_amd", "csrc")
    for f in sorted(os.listdir(d)) + ["../../include/mcbs.h"]:
        if f.endswith((".hip", ".h")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()


def git_head() -> str:
    try:
        return subprocess.run(["git", "-C", REPO, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
    except Exception:
        return ""
