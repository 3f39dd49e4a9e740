#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X step engine.

Metric (BASELINE.json): env-steps/sec at 65536 envs, CyberBattleChain-10; bit-exact vs CPU ref.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (`mcbs_step`, one fused kernel launch = CyberBattleEnv.step for every
env of the rank's shard) over one batch of actions that is already resident in HBM.  Workload per GPU:
CyberBattleChain size=10, 65 536 envs, attacker only, goal own 100 %, auto-reset, episodes truncated at 2 000
steps (marlon's max_timesteps, attack_wrapper.py:38); actions are valid random actions in the style of
CyberBattleEnv.sample_valid_action, recorded by an untimed rollout of the same engine and replayed from an
HBM ring (the replay is exact: attacker-only Chain has no randomness).  Weak scaling: every rank owns its own
65 536 envs (global env ids rank*E ..), no collective on the data path; the only collectives are the timing
barrier / MAX and an optional all_gather of episode returns after the timed region.

The timed region replays the K steps from a hipGraph (launch-bound inner loop captured once), bracketed by
barrier + synchronize on both sides.  The dominant kernel's average launch duration is measured in the same process
with HIP events on the launch stream bracketing that timed region (/ K; a per-launch event-pair figure from an eager
replay of the same K steps is printed beside it as an upper bound) and reported as a fraction of the HBM roofline;
the CPU oracle (oracle/, a port of the reference's algorithm, NOT the product) is timed on one host core on a
bounded sample of the same action ring, and its rewards are compared with the GPU's while at it.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

ENVS_PER_GPU = 65536
B_STEP = 348            # algorithmic bytes per env-step, SURVEY.md section 8(d) (attacker-only tier)
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def build_engine(rank: int, local_rank: int, n_envs: int, max_episode_steps: int):
    from marlon_amd import engine, flatten
    from marlon_amd._abi import EnvSpec
    from marlon_amd.samples import chainpattern

    topo = flatten.flatten(chainpattern.new_environment(10))
    spec = EnvSpec(n_envs=n_envs, maximum_node_count=12, maximum_total_credentials=12,
                   attacker_goal=dict(own_atleast_percent=1.0), auto_reset=True,
                   max_episode_steps=max_episode_steps, seed=12345, env_id_base=rank * n_envs, device=local_rank)
    return engine.BatchEngine(topo, spec, device=f"cuda:{local_rank}"), topo, spec


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--max-episode-steps", type=int, default=2000)
    ap.add_argument("--cpu-envs", type=int, default=65536)
    ap.add_argument("--no-graph", action="store_true", help="time eager launches instead of a hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to rehearse the "
                    "multi-rank code path on a single-GPU box together with --single-device)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
            return 2
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    backend = args.backend
    if world > 1:
        if backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
                probe = torch.zeros(1, device=f"cuda:{local_rank}")
                dist.all_reduce(probe)                      # RCCL communicators are created lazily: fail here, not in the timed region
                torch.cuda.synchronize()
            except Exception as exc:                        # the data path has no collective: only the timing barrier / MAX need one,
                print(f"bench.py: RCCL unavailable ({exc}); timing collectives over gloo", file=sys.stderr)   # so gloo is enough
                if dist.is_initialized():
                    dist.destroy_process_group()
                backend = "gloo"
                dist.init_process_group("gloo")
        else:
            dist.init_process_group(backend)
    coll_dev = torch.device(f"cuda:{local_rank}") if backend == "nccl" else torch.device("cpu")

    def barrier():
        if world > 1:
            dist.barrier()

    E, K, W = args.envs_per_gpu, args.steps, args.warmup
    from marlon_amd._abi import EnvSpec  # noqa: F401
    eng, topo, spec = build_engine(rank, local_rank, E, args.max_episode_steps)
    dev = eng.device

    # ---- untimed: record W+K batches of valid random actions into an HBM ring, then rewind ----
    ring = torch.empty((W + K, E, 5), dtype=torch.int32, device=dev)
    for t in range(W + K):
        eng.sample_actions(True, seed=12345, step=t, out=ring[t])
        eng.step(ring[t], with_info=False)
    torch.cuda.synchronize()
    eng.reset()          # back to the initial state: the replay below repeats the recorded trajectory exactly
    rewards = torch.empty((K, E), dtype=torch.float32, device=dev)
    dones = torch.empty((K, E), dtype=torch.uint8, device=dev)
    lib, h = eng.lib, eng._h

    def launch(t_ring: int, t_out: int, stream: int) -> None:
        rc = lib.mcbs_step(h, ring[t_ring].data_ptr(), rewards[t_out].data_ptr(), dones[t_out].data_ptr(), None, stream)
        if rc != 0:
            raise RuntimeError(lib.mcbs_last_error().decode())

    # ---- warm-up (untimed) ----
    st = torch.cuda.current_stream().cuda_stream
    for t in range(W):
        launch(t, t % K, st)
    torch.cuda.synchronize()

    graph = None
    if not args.no_graph:
        # the K timed steps captured once into a hipGraph (ring / output addresses are fixed per step)
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                s = torch.cuda.current_stream().cuda_stream
                for t in range(K):
                    launch(W + t, t, s)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # graph capture does not execute: state is still "after warm-up"

    # ---- timed region: exactly K steps ----
    barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()         # HIP events on the launch stream (torch's current stream IS the stream the K launches go to)
    if graph is not None:
        graph.replay()
    else:
        for t in range(K):
            launch(W + t, t, st)
    ev1.record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    region_us = ev0.elapsed_time(ev1) * 1e3 / K          # device time per launch over the timed region, launch gaps included
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    reward_sum_timed = rewards.double().sum(dim=0)          # per-env return over the K timed steps
    n_done = int(dones.sum().item())

    # ---- cross-check: one HIP event pair around EACH launch, same K steps, eager, from the same start state.  The pair
    # itself costs ~2 us per launch at this kernel size, so this figure is an upper bound; `kernel_us` (the roofline's
    # denominator) is the event-bracketed timed region / K, which is what rocprofv3's kernel trace agrees with ----
    eng.reset()
    for t in range(W):
        launch(t, t % K, st)
    torch.cuda.synchronize()
    eng.timing_enable(True)
    for t in range(K):
        launch(W + t, t, st)
    kernel_ms, launches = eng.timing_read()
    eng.timing_enable(False)
    pair_us = kernel_ms * 1e3 / max(1, launches)
    kernel_us = region_us
    same = bool(torch.equal(rewards.double().sum(dim=0), reward_sum_timed))

    # ---- extra, NOT the headline: the same K recorded steps through mcbs_step_many (one launch, no per-step launch cost) ----
    eng.reset()
    for t in range(W):
        launch(t, t % K, st)
    many_r = torch.empty((K, E), dtype=torch.float32, device=dev)
    many_d = torch.empty((K, E), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    m0.record()
    eng.step_many(ring[W:W + K], many_r, many_d)
    m1.record()
    torch.cuda.synchronize()
    many_us = m0.elapsed_time(m1) * 1e3 / K
    many_same = bool(torch.equal(many_r.double().sum(dim=0), reward_sum_timed))

    # ---- optional logging collective (not on the data path): episode returns of every rank ----
    if world > 1:
        mine = reward_sum_timed.to(coll_dev)
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)

    result = None
    if rank == 0:
        bytes_per_launch = float(B_STEP) * E
        achieved = bytes_per_launch / (kernel_us * 1e-6) / 1e9
        traffic = None
        tf = os.path.join(REPO, "profiles", "traffic_step_kernel.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get("bytes_per_launch")
            except Exception:
                traffic = None
        result = {
            "metric": "env-steps/sec at 65536 envs, CyberBattleChain-10; bit-exact vs CPU ref",
            "value": world * E * K / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed * 1e3 / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": f"CyberBattleChain size=10, {E} envs per GPU, attacker-only, recorded valid random actions, "
                                   f"auto-reset, truncation at {args.max_episode_steps} steps",
                       "envs_per_gpu": E, "launch": "hipGraph replay" if graph is not None else "eager",
                       "episodes_ended_in_timed_region_rank0": n_done},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "mcbs::step_kernel<0, 0, true, 0, false>", "kernel_us": kernel_us, "launches_timed": K,
                         "kernel_us_event_pair_per_launch": pair_us,
                         "algorithmic_bytes_per_launch": bytes_per_launch, "bytes_per_env_step": B_STEP,
                         "replay_rewards_equal_timed_region": same},
            # scripted-sequence entry point (no reference counterpart): K steps in ONE launch; reported beside, never as, `value`
            "step_many": {"us_per_step": many_us, "env_steps_per_s_rank0": E / (many_us * 1e-6), "rewards_equal_timed_region": many_same},
        }
        if not args.no_cpu_baseline:
            n = min(args.cpu_envs, E)
            acts = ring[W:W + K, :n].cpu().numpy()
            # the oracle starts from reset; the GPU's timed region started after W warm-up steps: feed it those too
            warm = ring[:W, :n].cpu().numpy()
            import numpy as _np
            full = _np.concatenate([warm, acts], axis=0)
            from oracle.oracle import Oracle
            import copy
            s2 = copy.copy(spec)
            s2.n_envs = n
            ref_sum = reward_sum_timed[:n].cpu().numpy()

            def cpu_leg(threads: int):
                orc = Oracle(topo, s2)
                orc.run(full[:W], threads)                       # warm-up steps, untimed
                t0 = time.perf_counter()
                tot = orc.run(full[W:], threads)                 # K steps of every env inside C (env-major: cache-resident state)
                dt = time.perf_counter() - t0
                del orc
                return dt, bool(_np.array_equal(tot, ref_sum))

            dt, eq = cpu_leg(1)
            result["cpu_baseline"] = {
                "value": n * K / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
                "sample": f"first {n} envs x {K} steps of the same action ring ({dt:.1f} s), scalar C oracle oracle/cbs_oracle.c, "
                          f"K-step loop inside C",
                "rewards_equal_gpu": eq,
            }
            cores = max(1, min(os.cpu_count() or 1, 64, n))
            dt_all, eq_all = cpu_leg(cores)      # envs never interact: disjoint env ranges on all host cores
            result["cpu_baseline"]["all_cores"] = {"value": n * K / dt_all, "cores": cores, "seconds": round(dt_all, 2),
                                                   "rewards_equal_gpu": eq_all}
        print(json.dumps(result))
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
