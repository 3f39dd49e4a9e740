#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X step engine.

Metric (BASELINE.json): env-steps/sec at 65536 envs, CyberBattleChain-10; bit-exact vs CPU ref.

    python bench.py --gpus N --steps K --warmup W            (N > 1: starts N ranks itself, see below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (`mcbs_step`, one fused kernel launch = CyberBattleEnv.step for every
env of the rank's shard) over one batch of actions that is already resident in HBM.  Workload per GPU:
CyberBattleChain size=10, 65 536 envs, attacker only, goal own 100 %, auto-reset, episodes truncated at 2 000
steps (marlon's max_timesteps, attack_wrapper.py:38); actions are valid random actions in the style of
CyberBattleEnv.sample_valid_action, recorded by an untimed rollout of the same engine and replayed from an
HBM ring (the replay is exact: attacker-only Chain has no randomness).  Weak scaling: every rank owns its own
65 536 envs (global env ids rank*E ..), no collective on the data path; the only collectives are the timing
barrier / MAX / gathers and an all_gather of episode returns after the timed region.

The timed region replays the K steps from a hipGraph (launch-bound inner loop captured once), bracketed by
barrier + synchronize on both sides (barrier, synchronize, K steps, synchronize, barrier; every rank times its own K steps between the two
synchronizes and the job's time is the MAX over ranks).  The W warm-up steps are a second captured graph replayed right before.  Before
that, untimed, both graphs are replayed once as a REHEARSAL (their first launch uploads them; it also touches every page the steps use) and
the engine is put back to its start state: the driver's `--steps 20` region is 0.1 ms of device work, so a first-launch cost of a few
microseconds would otherwise be a tenth of it.  The dominant kernel's average launch duration is measured in the same process with HIP
events on the launch stream bracketing that timed region (/ K) and reported as a fraction of the HBM roofline; every launch of that kernel
this process makes is a graph replay of the same shape (the recording rollout runs as ONE mcbs_rollout_random launch — a different kernel —
and the per-launch event-pair cross-check is behind --event-pairs), so `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-extras
--no-cpu-baseline` reports the same regime (profiles/round3_bench_headline*.json).
Beside the headline (never as `value`): `configs` — the step kernels of BASELINE.json's configs 3, 4 (one GPU's shard) and
5 (one GPU's shard) timed the same way, with their roofline fraction from MEASURED HBM bytes (profiles/round2_step_*.json,
tied to the kernel sources by a hash); `observe` — the observation tier; `cpu_baseline` — the CPU oracle (oracle/, a port of
the reference's algorithm, NOT the product) timed on the host cores on a bounded sample of the same action ring, its rewards
and termination flags compared with the GPU's, and the reference's own Python path as timed in the build container.

`python bench.py --gpus N` invoked plainly (no RANK / WORLD_SIZE in the environment) starts the N ranks itself: before
anything touches the GPU it runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...` on this same file as a
child process, passes rank 0's JSON line through and exits with the child's code.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

ENVS_PER_GPU = 65536
B_STEP = 348            # algorithmic bytes per env-step, SURVEY.md section 8(d) (attacker-only tier)
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
METRIC = "env-steps/sec at 65536 envs, CyberBattleChain-10; bit-exact vs CPU ref"
HEADLINE_KERNEL = "mcbs::step_kernel<0, 0, false, 0>"   # <PHASE whole step, packed sets, hot image through L1/L2, no defender>


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--max-episode-steps", type=int, default=2000)
    ap.add_argument("--cpu-envs", type=int, default=65536)
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="CPU oracle sample: repeat the recorded steps until about this long (1 core; half of it on all cores)")
    ap.add_argument("--no-graph", action="store_true", help="time eager launches instead of a hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--event-pairs", action="store_true", help="also replay the K steps eagerly with one HIP event pair per launch (cross-check; "
                    "adds eager launches of the headline kernel to the process: leave off under rocprofv3)")
    ap.add_argument("--no-extras", action="store_true", help="skip the `configs` and `observe` legs (they run at N = 1 only)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend of the timing collectives for N>1 (nccl = RCCL; gloo is the "
                    "agreed fallback when RCCL cannot be brought up on every rank)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal on a one-GPU box: every rank uses cuda:0")
    ap.add_argument("--rehearse", action="store_true", help="launcher / collective / JSON plumbing only, NO engine and no GPU (CPU-safe: "
                    "tests/test_distributed_gloo.py); prints value null")
    ap.add_argument("--master-port", type=int, default=0, help="self-launch only: rendezvous port (0 = pick a free one)")
    return ap.parse_args(argv)


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child `torch.distributed.run` (never re-exec: this process
    has not touched the GPU and will not), forward rank 0's JSON line, return the child's exit code."""
    port = args.master_port or free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + [a for a in sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this pool
    env["MCBS_BENCH_SELF_LAUNCHED"] = "1"
    print(f"bench.py: starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line)
    if proc.returncode != 0:
        print(f"bench.py: a rank failed (torch.distributed.run exit code {proc.returncode})", file=sys.stderr)
        return proc.returncode
    return 0 if line is not None else 1


def init_collectives(world: int, rank: int, local_rank: int, want: str, use_gpu: bool):
    """Process group for the timing collectives.  The default group is always gloo (it cannot fail half-way); RCCL is brought up as
    a second group inside a try, every rank reports whether its probe all-reduce went through, and the ranks AGREE (MIN over gloo)
    before any of them uses it — a rank whose RCCL init failed can therefore not leave the others waiting at an RCCL barrier.
    Returns (backend name, group or None, device for collective tensors)."""
    import datetime
    import torch
    import torch.distributed as dist
    if world == 1:
        return "none", None, torch.device("cpu")
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=300))
    if want != "nccl" or not use_gpu:
        return "gloo", None, torch.device("cpu")
    ok, group = 1, None
    try:
        group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120))
        probe = torch.ones(1, device=f"cuda:{local_rank}")
        dist.all_reduce(probe, group=group)                 # communicators are created lazily: fail here, not in the timed region
        torch.cuda.synchronize()
        ok = int(probe.item() == world)
    except Exception as exc:
        print(f"bench.py rank {rank}: RCCL unavailable ({type(exc).__name__}: {exc})", file=sys.stderr)
        ok = 0
    flag = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)             # gloo: every rank takes the same decision
    if int(flag.item()) == 1:
        return "nccl", group, torch.device(f"cuda:{local_rank}")
    if rank == 0:
        print("bench.py: RCCL did not come up on every rank; timing collectives over gloo (the data path has no collective)", file=sys.stderr)
    return "gloo", None, torch.device("cpu")


def load_traffic(name: str, kernel_prefix: str):
    """Measured HBM bytes per launch of one kernel from profiles/round3_<name>.json (else round2_) (tools/profile_all.sh + tools/pmc_summary.py).
    Returns (bytes_per_launch or None, info dict).  A file taken on other kernel sources than the ones this tree builds is STALE:
    its figure is not used."""
    from tools import workloads as W
    path = next((q for q in (os.path.join(REPO, "profiles", f"{r}_{name}.json") for r in ("round3", "round2")) if os.path.exists(q)), None)
    if path is None:
        return None, {"file": f"profiles/round3_{name}.json", "status": "missing"}
    info = {"file": os.path.relpath(path, REPO)}
    try:
        d = json.load(open(path))
        k = next(k for k in d["kernels"] if k["kernel"].startswith(kernel_prefix) and k.get("hbm_bytes_per_launch") is not None)
    except Exception as exc:
        info["status"] = f"unreadable ({type(exc).__name__})"
        return None, info
    info.update(kernel=k["kernel"], rocprof_avg_us=k["avg_us"], rocprof_min_us=k["min_us"], calls=k["calls"],
                csrc_sha256=d.get("csrc_sha256", "")[:16], git_head=d.get("git_head_when_collected", ""))
    if d.get("csrc_sha256") != W.csrc_sha256():
        info["status"] = "stale: the kernel sources changed since these counters were taken"
        return None, info
    info["status"] = "current"
    return float(k["hbm_bytes_per_launch"]), info


def reference_python_timing():
    """The reference's own CPU path, timed in the build container (it cannot travel to the GPU box)."""
    try:
        d = json.load(open(os.path.join(REPO, "profiles", "reference_cpu_timing.json")))
        one = next(r for r in d["rows"] if r["config"] == "chain10" and r["processes"] == 1)
        many = next(r for r in d["rows"] if r["config"] == "chain10" and r["processes"] > 1)
        return {"value": one["step_only_steps_per_s"], "unit": "env-steps/s", "cores": 1, "kind": "reference",
                "all_cores": {"value": many["step_only_steps_per_s_sum"], "cores": many["processes"]},
                "whole_loop_value": one["loop_steps_per_s"],
                "host": "build container, 8 cores (NOT the GPU box: the reference cannot travel)", "source": "profiles/reference_cpu_timing.json",
                "what": "unmodified CyberBattleEnv.step, Chain-10 bounds 12/12, attacker only, one env per process",
                "note": d.get("note", "")}
    except Exception as exc:
        return {"value": None, "error": f"{type(exc).__name__}: {exc}"}


def rehearse(args, world, rank, local_rank) -> int:
    """No engine, no GPU: the launcher, the agreed-backend collectives and the JSON assembly with a stand-in timed region."""
    import torch
    import torch.distributed as dist
    backend, group, coll_dev = init_collectives(world, rank, local_rank, args.backend, use_gpu=False)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [elapsed]
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        allt = [torch.empty_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        per_rank = [float(x.item()) for x in allt]
        mine = torch.full((4,), float(rank), dtype=torch.float64)
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        assert [float(g[0]) for g in gathered] == [float(r) for r in range(world)]
    # the `configs` legs of a multi-rank run (configs 4 and 5: every rank its shard): one barrier-bracketed region and one gather of
    # times per configuration, in the same order on every rank
    cfg_rows = []
    for i, name in enumerate(("config4", "config5") if world > 1 else ("config3", "config4", "config5")):
        if world > 1:
            dist.barrier()
        c0 = time.perf_counter()
        time.sleep(0.002 * (rank + 1))
        el = time.perf_counter() - c0
        if world > 1:
            dist.barrier()
            tt = torch.tensor([el], dtype=torch.float64)
            allt = [torch.empty_like(tt) for _ in range(world)]
            dist.all_gather(allt, tt)
            el = max(float(x.item()) for x in allt)
        cfg_rows.append({"name": name, "n_gpus": world, "ms_host_clock_max_over_ranks": el * 1e3, "env_steps_per_s": None})
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": None, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "configs": cfg_rows,
                          "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
                          "data": "synthetic", "rehearsal": "launcher and collectives only: no engine, no GPU, nothing measured",
                          "collective_backend": backend, "ranks_in_group": dist.get_world_size() if world > 1 else 1,
                          "per_rank_ms": [x * 1e3 for x in per_rank],
                          "self_launched": os.environ.get("MCBS_BENCH_SELF_LAUNCHED") == "1"}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main() -> int:
    args = parse_args()
    have_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not have_launcher:
        return self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks", file=sys.stderr)
        return 2
    if args.rehearse:
        return rehearse(args, world, rank, local_rank)

    import numpy as np
    import torch
    import torch.distributed as dist
    from tools import workloads as W

    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    backend, group, coll_dev = init_collectives(world, rank, local_rank, args.backend, use_gpu=True)

    def barrier():
        if world > 1:
            dist.barrier(group=group)

    E, K, Wm = args.envs_per_gpu, args.steps, args.warmup

    # ---- headline: 65 536 Chain-10 envs per GPU, attacker only ----
    eng, topo, spec, desc = W.make_engine("headline", n_envs=E, env_id_base=rank * E, device=f"cuda:{local_rank}",
                                          max_episode_steps=args.max_episode_steps, seed=12345)
    # untimed: W+K batches of valid random actions recorded into an HBM ring by ONE launch of the looping step kernel (the random agent
    # sampled inside it: the actions mcbs_sample_actions(valid, seed, t) + mcbs_step would give, tests/test_gpu_parity.py)
    ring = eng.rollout_random(Wm + K, valid=True, seed=12345, first_step=0, record_actions=True)[2]
    torch.cuda.synchronize()
    leg = W.timed_leg(eng, ring, Wm, K, graph=not args.no_graph, barrier=barrier, restore=eng.rewind)
    elapsed_mine, region_us, rewards, dones = leg["elapsed_s"], leg["region_us"], leg["rewards"], leg["dones"]
    elapsed, per_rank = gather_times(elapsed_mine, world, coll_dev, group)
    reward_sum_timed = rewards.double().sum(dim=0)          # per-env return over the K timed steps
    done_cnt_timed = dones.long().sum(dim=0)                # per-env `terminated` flags raised in the K timed steps
    n_done = int(done_cnt_timed.sum().item())
    same = leg["rehearsal_equal"]
    kernel_us = region_us
    lib, h, dev = eng.lib, eng._h, eng.device
    st = torch.cuda.current_stream().cuda_stream

    def launch(t_ring: int, t_out: int, stream: int) -> None:
        rc = lib.mcbs_step(h, ring[t_ring].data_ptr(), rewards[t_out].data_ptr(), dones[t_out].data_ptr(), None, stream)
        if rc != 0:
            raise RuntimeError(lib.mcbs_last_error().decode())

    pair_us = None
    if args.event_pairs:
        # cross-check (off by default: its K eager launches of the same kernel would mix a second regime into a rocprofv3 trace of
        # this process): one HIP event pair around EACH launch, same K steps, eager, from the same start state.  The pair itself
        # costs ~2 us per launch at this kernel size, so this figure is an upper bound
        eng.reset()
        for t in range(Wm):
            launch(t, t % K, st)
        torch.cuda.synchronize()
        eng.timing_enable(True)
        for t in range(K):
            launch(Wm + t, t, st)
        kernel_ms, launches = eng.timing_read()
        eng.timing_enable(False)
        pair_us = kernel_ms * 1e3 / max(1, launches)
        same = same and bool(torch.equal(rewards.double().sum(dim=0), reward_sum_timed) and torch.equal(dones.long().sum(dim=0), done_cnt_timed))

    # ---- extra, NOT the headline: the same K recorded steps through mcbs_step_many (one launch, no per-step launch cost) ----
    eng.reset()
    if Wm:
        eng.step_many(ring[:Wm])
    many_r = torch.empty((K, E), dtype=torch.float32, device=dev)
    many_d = torch.empty((K, E), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    m0.record()
    eng.step_many(ring[Wm:Wm + K], many_r, many_d)
    m1.record()
    torch.cuda.synchronize()
    many_us = m0.elapsed_time(m1) * 1e3 / K
    many_same = bool(torch.equal(many_r.double().sum(dim=0), reward_sum_timed) and torch.equal(many_d.long().sum(dim=0), done_cnt_timed))

    # ---- optional logging collective (not on the data path): episode returns of every rank ----
    if world > 1:
        mine = reward_sum_timed.to(coll_dev)
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine, group=group)

    result = None
    parity_ok = same and many_same
    if rank == 0:
        bytes_per_launch = float(B_STEP) * E
        achieved = bytes_per_launch / (kernel_us * 1e-6) / 1e9
        traffic, tinfo = load_traffic("step_headline", W.STEP_KERNEL["headline"])
        if traffic is not None and E != ENVS_PER_GPU:
            traffic, tinfo["status"] = None, "not applicable: counters were taken at 65 536 envs per launch"
        binding = load_binding("step_headline", W.STEP_KERNEL["headline"])
        layout_b = layout_bytes_per_env_step(topo.n_nodes)
        per_rank_value = [E * K / x for x in per_rank]
        result = {
            "metric": METRIC,
            "value": world * E * K / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": Wm,
            "ms_per_step": elapsed * 1e3 / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": f"CyberBattleChain size=10, {E} envs per GPU, attacker-only, recorded valid random actions, "
                                   f"auto-reset, truncation at {args.max_episode_steps} steps",
                       "envs_per_gpu": E, "launch": "hipGraph replay" if not args.no_graph else "eager",
                       "episodes_ended_in_timed_region_rank0": n_done,
                       # where the host clock's region goes: the device's share (HIP events) and the host's calls around it
                       "timed_region_breakdown_us_rank0": leg["host_breakdown"]},
            "collective_backend": backend, "ranks_in_group": world,
            "per_rank_ms_per_step": [x * 1e3 / K for x in per_rank],
            "n1_value_hint": sum(per_rank_value) / len(per_rank_value),     # what ONE GPU did in this run (mean over ranks): compare with the N = 1 line
            "self_launched": os.environ.get("MCBS_BENCH_SELF_LAUNCHED") == "1",
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tinfo,
                         "kernel": HEADLINE_KERNEL, "kernel_us": kernel_us, "launches_timed": K,
                         "kernel_us_event_pair_per_launch": pair_us,
                         # the committed rocprofv3 kernel trace of THIS command (every launch of the kernel a graph replay of this shape):
                         # its average duration and the fraction it gives; with the profiler attached bench.py's own kernel_us was ...
                         "rocprof": load_rocprof("bench_headline", W.STEP_KERNEL["headline"], bytes_per_launch),
                         "algorithmic_bytes_per_launch": bytes_per_launch, "bytes_per_env_step": B_STEP,
                         # what the packed layout itself loads and stores per env-step (the contract's 348 B model charges 64-byte node
                         # rows and a 32-byte header this layout does not have): DESIGN.md section 4
                         "layout_bytes_per_env_step": layout_b,
                         "frac_layout": layout_b * E / (kernel_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "frac_of_measured_traffic": None if traffic is None else traffic / (kernel_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         # what actually binds the launch: one wavefront per SIMD, its cycles mostly parked on memory (SQ counters of
                         # the committed profile): latency, not bandwidth — the 10 MB of state are L2 / Infinity-Cache resident
                         "binding": binding,
                         "replay_rewards_and_dones_equal_timed_region": same},
            # scripted-sequence entry point (no reference counterpart): K steps in ONE launch; reported beside, never as, `value`.
            # Under the contract's 348 B model it exceeds the HBM peak (frac > 1): the state never leaves the caches between steps
            "step_many": {"us_per_step": many_us, "env_steps_per_s_rank0": E / (many_us * 1e-6), "rewards_and_dones_equal_timed_region": many_same,
                          "frac_by_348B_model": B_STEP * E / (many_us * 1e-6) / 1e9 / HBM_PEAK_GBS},
        }
    ring_cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        n = min(args.cpu_envs, E)
        ring_cpu = ring[:, :n].cpu().numpy()
        ref_sum, ref_done = reward_sum_timed[:n].cpu().numpy(), done_cnt_timed[:n].cpu().numpy().astype(np.int32)
    eng.close()
    del ring, rewards, dones, many_r, many_d, leg
    torch.cuda.empty_cache()

    # ---- beside the headline, every rank: the step kernels of the other BASELINE.json configurations.  N = 1: configs 3, 4 (one GPU's
    # shard) and 5 (one GPU's shard); N > 1: configs 4 and 5 are DEFINED as shards over the GPUs of a node, so every rank runs its own
    # shard (global env ids rank * shard ..) inside the same barrier / synchronize / MAX-over-ranks bracket and rank 0 reports the
    # aggregate env-steps/s ----
    if not args.no_extras:
        try:
            cfgs = extras_configs(W, world, rank, local_rank, barrier, lambda x: gather_times(x, world, coll_dev, group)[0])
            if rank == 0:
                result["configs"] = cfgs
        except Exception as exc:      # the headline stands on its own; an extras failure is reported, not hidden
            if rank == 0:
                result["extras_error"] = f"configs: {type(exc).__name__}: {exc}"
            parity_ok = False
    # ---- rank 0 only (the other ranks wait at the closing barrier): the headline kernel with episodes ending inside the timed
    # region, the observation tier and the wrapper tier ----
    if rank == 0 and not args.no_extras:
        try:
            result["headline_with_resets"] = extras_resets(W, E, K if K <= 1000 else 1000, Wm, f"cuda:{local_rank}", not args.no_graph)
            parity_ok = parity_ok and result["headline_with_resets"]["rewards_dones_and_episodes_equal_rehearsal"]
            if world == 1:
                result["batch_sweep"] = extras_batch_sweep(W)
                result["observe"] = extras_observe(W)
                result["wrapper"] = extras_wrapper()
        except Exception as exc:
            result["extras_error"] = (result.get("extras_error", "") + f" rank-0 extras: {type(exc).__name__}: {exc}").strip()
            parity_ok = False

    if rank == 0 and ring_cpu is not None:
        from oracle.oracle import Oracle
        import copy
        n = ring_cpu.shape[1]
        s2 = copy.copy(spec)
        s2.n_envs = n

        def cpu_leg(threads: int, budget_s: float):
            # the oracle starts from reset like the GPU did: W warm-up steps (untimed), then the K timed ones; the whole pass is
            # repeated from a fresh oracle until the time budget is used, every pass checked against the GPU's results
            steps_done, dt, eq = 0, 0.0, True
            while dt < budget_s:
                orc = Oracle(topo, s2)
                if Wm:
                    orc.run(ring_cpu[:Wm], threads)
                c0 = time.perf_counter()
                tot, ended = orc.run(ring_cpu[Wm:], threads, count_terminated=True)
                dt += time.perf_counter() - c0
                steps_done += n * K
                eq = eq and bool(np.array_equal(tot, ref_sum)) and bool(np.array_equal(ended, ref_done))
                del orc
            return steps_done / dt, dt, eq, steps_done

        v1, dt1, eq1, sd1 = cpu_leg(1, args.cpu_seconds)
        cores = max(1, min(os.cpu_count() or 1, 64, n))
        va, dta, eqa, sda = cpu_leg(cores, args.cpu_seconds / 2)     # envs never interact: disjoint env ranges on all host cores
        result["cpu_baseline"] = {
            "value": v1, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"first {n} envs x {K} steps of the same action ring, repeated {sd1 // (n * K)}x from reset ({dt1:.1f} s of CPU work), "
                      f"scalar C oracle oracle/cbs_oracle.c, K-step loop inside C (env-major: each env's state stays in cache)",
            "rewards_and_dones_equal_gpu": eq1,
            "all_cores": {"value": va, "cores": cores, "seconds": round(dta, 2), "rewards_and_dones_equal_gpu": eqa},
            "reference_python": reference_python_timing(),
        }
        parity_ok = parity_ok and eq1 and eqa

    if rank == 0:
        result["parity_ok"] = bool(parity_ok)
        if not parity_ok:
            # "bit-exact vs CPU ref" is part of the metric: a run whose checks fail has no headline value
            result["value_unverified"] = result["value"]
            result["value"] = None
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if parity_ok else 3


def gather_times(elapsed_mine: float, world: int, coll_dev, group):
    """(MAX over ranks, every rank's time) of one timed region."""
    if world == 1:
        return elapsed_mine, [elapsed_mine]
    import torch
    import torch.distributed as dist
    tt = torch.tensor([elapsed_mine], dtype=torch.float64, device=coll_dev)
    allt = [torch.empty_like(tt) for _ in range(world)]
    dist.all_gather(allt, tt, group=group)
    per_rank = [float(x.item()) for x in allt]
    return max(per_rank), per_rank


def max_over_ranks(x: float, world: int, coll_dev, group) -> float:
    return gather_times(x, world, coll_dev, group)[0]


def layout_bytes_per_env_step(n_nodes: int) -> int:
    """Bytes one env-step of the PACKED layout loads and stores (marlon_amd/csrc/mcbs_step.hip, level-1 loads and the store round;
    the shared hot topology image is L1-resident and not counted).  Loads: header uint4 16 + action row 20 + discovery
    order head 16 + credential cache head 32 + every 4-byte node row (48 for <= 12 nodes, else 64) + the eight sets as one uint4 16 +
    {cum_reward, availability} 16.  Stores: the target's row 4 + sets 16 + header 16 + {cum_reward, availability} 16 + reward 4 +
    terminated 1 (+ 1 B / 2 B per newly listed node / credential: < 1 per step on average, not counted)."""
    return (16 + 20 + 16 + 32 + (48 if n_nodes <= 12 else 64) + 16 + 16) + (4 + 16 + 16 + 16 + 4 + 1)


def load_rocprof(name: str, kernel_prefix: str, bytes_per_launch: float):
    """Average duration of the kernel in the committed `rocprofv3 --kernel-trace --stats` summary of bench.py itself, the roofline
    fraction that average gives, and what bench.py's own event bracket measured in that profiled run (the profiler's per-dispatch
    time-stamping lengthens every launch a little, so the two figures of ONE run are the pair to compare)."""
    path = os.path.join(REPO, "profiles", f"round3_{name}.json")
    try:
        d = json.load(open(path))
        k = next(k for k in d["kernels"] if k["kernel"].startswith(kernel_prefix))
    except Exception as exc:
        return {"file": os.path.relpath(path, REPO), "status": f"missing or unreadable ({type(exc).__name__})"}
    run = d.get("run", {})
    return {"file": os.path.relpath(path, REPO), "command": d.get("command"), "calls": k["calls"], "avg_us": k["avg_us"], "min_us": k["min_us"],
            "max_us": k["max_us"], "median_us": k.get("median_us"), "frac_by_avg_us": bytes_per_launch / (k["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "frac_by_median_us": None if not k.get("median_us") else bytes_per_launch / (k["median_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "kernel_us_of_the_profiled_run": (run.get("roofline") or {}).get("kernel_us"), "ms_per_step_of_the_profiled_run": run.get("ms_per_step"),
            "csrc_matches_this_tree": d.get("csrc_sha256") == __import__("tools.workloads", fromlist=["x"]).csrc_sha256()}


def load_binding(name: str, kernel_prefix: str):
    """What binds the kernel, from the SQ counters of the committed profile (any round: the counters describe the kernel's structure)."""
    for rnd in ("round3", "round2"):
        path = os.path.join(REPO, "profiles", f"{rnd}_{name}.json")
        try:
            d = json.load(open(path))
            k = next(k for k in d["kernels"] if k["kernel"].startswith(kernel_prefix) and k.get("wave_cycle_split"))
        except Exception:
            continue
        w = k["wave_cycle_split"]
        return {"what": "latency: one wavefront per SIMD, the launch lasts as long as one wavefront's chain of dependent memory accesses",
                "waves_per_launch": k.get("waves_per_launch"), "simds": 1024,
                "wave_cycles_parked_at_waitcnt": w.get("parked_at_waitcnt_or_barrier"), "issuing": w.get("issuing"),
                "issue_stalled": w.get("issue_stalled"), "source": os.path.relpath(path, REPO)}
    return {"what": "latency: one wavefront per SIMD (no SQ-counter profile found under profiles/)"}


def extras_configs(W, world: int, rank: int, local_rank: int, barrier, max_over):
    """Step kernels of BASELINE.json configs 3-5 (per-GPU shard for the 8-GPU ones), timed exactly like the headline (timed_leg: recorded
    valid actions, rehearsed hipGraph replay, barrier + synchronize on both sides, HIP events on the launch stream, MAX over ranks).
    N = 1: configs 3, 4 (one shard), 5 (one shard).  N > 1: configs 4 and 5, rank r running the shard of global env ids r * shard ..;
    `env_steps_per_s` is then the aggregate over the N shards.  Roofline fraction from the MEASURED HBM bytes of the same kernel (two
    --pmc passes, profiles/round*_step_<config>.json) — SURVEY 8(d)'s N*64/scan_frequency defender term is not used: the bit-mask /
    ring state never moves the per-node rows it charges.  Every rank returns; rank 0's list is the one reported."""
    import torch
    out = []
    names = (("config3", 300), ("config4", 300), ("config5", 300)) if world == 1 else (("config4", 300), ("config5", 300))
    for name, K in names:
        _, shard, _, _ = W.workload(name)
        base = rank * shard
        ring = W.record_ring(name, K, env_id_base=base, device=f"cuda:{local_rank}")
        eng, topo, spec, desc = W.make_engine(name, env_id_base=base, device=f"cuda:{local_rank}")
        # (an engine in its creation state replays the recorded trajectory exactly, defender draws included: Philox is keyed by
        # (seed, global env id, episode, step); mcbs_rewind puts the episode counters back to 0 after the rehearsal)
        leg = W.timed_leg(eng, ring, 0, K, graph=True, barrier=barrier, restore=eng.rewind)
        elapsed = max_over(leg["elapsed_s"])
        us = leg["region_us"]
        rewards, dones = leg["rewards"], leg["dones"]
        traffic, tinfo = load_traffic(f"step_{name}", W.STEP_KERNEL[name])
        row = {"workload": f"{desc}, {eng.E} envs per GPU", "name": name, "envs_per_gpu": eng.E, "n_gpus": world, "nodes": topo.n_nodes, "steps": K,
               "us_per_step": us, "ms_per_step_host_clock_max_over_ranks": elapsed * 1e3 / K,
               "env_steps_per_s": world * eng.E * K / elapsed, "env_steps_per_s_by_events_rank0": eng.E / (us * 1e-6),
               "reward_sum_rank0": float(rewards.double().sum()), "episodes_ended_rank0": int(dones.sum()),
               "rehearsal_equal": leg["rehearsal_equal"], "kernel": tinfo.get("kernel"),
               "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": traffic, "traffic_source": tinfo,
                            "achieved": None if traffic is None else traffic / (us * 1e-6) / 1e9,
                            "frac": None if traffic is None else traffic / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                            "bytes_per_env_step_measured": None if traffic is None else traffic / eng.E}}
        out.append(row)
        eng.close()
        del ring, rewards, dones, leg
        torch.cuda.empty_cache()
    return out


def extras_resets(W, E: int, K: int, Wm: int, device: str, graph: bool):
    """The headline kernel with episodes ENDING inside the timed region.  The metric's step includes the auto-reset, but at
    marlon's truncation (2 000 steps) no episode ends within a thousand steps of a fresh batch.  Same workload and kernel with truncation
    at 100 steps and the envs' step counters staggered over 0..99 beforehand (mcbs_set_state), so that about 1 % of the envs are
    truncated and re-initialised by the wave-cooperative reset tail in EVERY step.  Reported beside `value`, never as it."""
    import numpy as np
    import torch
    trunc = 100
    eng, topo, spec, desc = W.make_engine("headline", n_envs=E, device=device, max_episode_steps=trunc, seed=12345)
    hdr, nodes, order, cache = eng.get_state()
    hdr = hdr.copy()
    hdr["step_count"] = (np.arange(E, dtype=np.int64) * 7919 % trunc).astype(hdr["step_count"].dtype)     # ages 0..99, scattered over the wavefronts

    def restore():
        eng.rewind()
        eng.set_state(hdr, nodes, order, cache)

    def episodes():
        return eng.get_state()[0]["episode"].astype(np.int64)

    restore()
    ring = eng.rollout_random(Wm + K, valid=True, seed=12345, first_step=0, record_actions=True)[2]
    torch.cuda.synchronize()
    restore()
    ep0 = episodes()
    leg = W.timed_leg(eng, ring, Wm, K, graph=graph, barrier=lambda: None, restore=restore)
    ended = int((episodes() - ep0).sum())               # resets since the restore before the timed pass: warm-up + timed steps
    us = leg["region_us"]
    # the same W+K steps through the looping kernel from the same start state: rewards, flags and episode counts must agree
    restore()
    if Wm:
        eng.step_many(ring[:Wm])
    mr, md = eng.step_many(ring[Wm:Wm + K])
    torch.cuda.synchronize()
    same = bool(leg["rehearsal_equal"] and torch.equal(mr, leg["rewards"]) and torch.equal(md, leg["dones"]) and int((episodes() - ep0).sum()) == ended)
    out = {"workload": f"{desc}, {E} envs, truncation at {trunc} steps, step counters staggered: ~{100 // trunc} % of the envs end and are reset in every step",
           "kernel": HEADLINE_KERNEL, "steps": K, "warmup": Wm, "us_per_step": us, "ms_per_step_host_clock": leg["elapsed_s"] * 1e3 / K,
           "env_steps_per_s": E * K / leg["elapsed_s"], "episodes_ended": ended, "episodes_ended_per_step": ended / max(1, Wm + K),
           "terminated_flags_in_timed_region": int(leg["dones"].sum()),
           "frac_by_348B_model": B_STEP * E / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
           "rewards_dones_and_episodes_equal_rehearsal": same}
    eng.close()
    return out


def extras_batch_sweep(W, sizes=(4096, 16384, 32768, 65536, 131072), K: int = 300):
    """The headline kernel at other batch sizes (same workload, hipGraph replay, HIP events): what part of a step is FIXED — the
    dependent-launch gap plus one wavefront's chain of memory accesses — and what part grows with the batch.  A least-squares line
    through the points gives `fixed_us` and `ps_per_env`; `marginal_GBps_layout` = the packed layout's 221 B per env-step over that
    slope: the rate at which additional envs are served once the fixed part is paid (state is L2 / Infinity-Cache resident up to
    131 072 envs, so this is not an HBM rate)."""
    import torch
    pts = []
    for n in sizes:
        ring = W.record_ring("headline", K, n_envs=n)
        eng, topo, spec, desc = W.make_engine("headline", n_envs=n)
        us, rewards, dones = W.graph_replay_us(eng, ring, K)
        pts.append({"envs": n, "us_per_step": us, "env_steps_per_s": n / (us * 1e-6)})
        eng.close()
        del ring, rewards, dones
        torch.cuda.empty_cache()
    xs, ys = [p["envs"] for p in pts], [p["us_per_step"] for p in pts]
    mx, my = sum(xs) / len(xs), sum(ys) / len(ys)
    slope = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs)
    layout = layout_bytes_per_env_step(12)
    return {"kernel": HEADLINE_KERNEL, "steps": K, "points": pts, "fixed_us": my - slope * mx, "ps_per_env": slope * 1e6,
            "marginal_GBps_layout": layout / (slope * 1e-6) / 1e9 if slope > 0 else None,
            "marginal_GBps_348B_model": B_STEP / (slope * 1e-6) / 1e9 if slope > 0 else None}


def extras_observe(W):
    """Observation tier (SURVEY 8(d) B_obs): the whole observation in the reference's dtypes / layout, one mcbs_observe per call."""
    import torch
    out = []
    for name, label in (("headline", "all fields"), ("config3", "all fields")):
        ring = W.record_ring(name, 40)
        eng, topo, spec, desc = W.make_engine(name)
        us, bpe, obs = W.observe_us(eng, ring, W.OBS_FIELDS, reps=20)
        gbps = bpe * eng.E / (us * 1e-6) / 1e9
        row = {"workload": f"{desc}, {eng.E} envs, {label} (int32 fields + three int8 action masks)", "name": name, "envs": eng.E,
               "us_per_observe": us, "bytes_per_env": bpe, "GBps": gbps, "frac": gbps / HBM_PEAK_GBS, "bound": "hbm (writes)",
               "env_observations_per_s": eng.E / (us * 1e-6)}
        traffic, tinfo = load_traffic(f"obs_{name}", "mcbs::obs_small_kernel")
        row["traffic"] = traffic
        row["traffic_source"] = tinfo
        out.append(row)
        del obs
        if name == "headline":      # the MaskablePPO path: small fields + the flat Discrete mask (connect | local | remote)
            us2, bpe2, obs2 = W.observe_us(eng, ring, W.OBS_FIELDS[:5] + ["mask_discrete"], reps=20, advance=0)
            g2 = bpe2 * eng.E / (us2 * 1e-6) / 1e9
            out.append({"workload": f"{desc}, {eng.E} envs, small fields + mask_discrete (MaskablePPO path)", "name": "headline_discrete",
                        "envs": eng.E, "us_per_observe": us2, "bytes_per_env": bpe2, "GBps": g2, "frac": g2 / HBM_PEAK_GBS, "bound": "hbm (writes)"})
            del obs2
            # the same path WITHOUT a materialised mask: small fields only, the Discrete mask applied to the policy's logits in place
            # (mcbs_mask_logits, rebuilt from the observation's 64-byte digest); write-only: masked-out logits are overwritten, the
            # allowed ones are neither read nor written
            us3, bpe3, obs3 = W.observe_us(eng, ring, W.OBS_FIELDS[:5], reps=20, advance=0)
            A = eng.discrete_action_count()
            logits = torch.zeros((eng.E, A), dtype=torch.float32, device=eng.device)
            eng.mask_logits(logits, fill=-1e8)
            written = 4.0 * float((logits != 0).sum()) / eng.E             # bytes per env the launch has to write: 4 per masked-out action
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                eng.mask_logits(logits)
            e1.record()
            torch.cuda.synchronize()
            us4 = e0.elapsed_time(e1) * 100.0
            g4 = written * eng.E / (us4 * 1e-6) / 1e9
            out.append({"workload": f"{desc}, {eng.E} envs, small fields only + mcbs_mask_logits on fp32 logits [E, {A}] (no mask materialised)",
                        "name": "headline_mask_logits", "envs": eng.E, "us_per_observe": us3, "bytes_per_env": bpe3,
                        "mask_logits_us": us4, "mask_logits_bytes_per_env": written, "logits_bytes_per_env": A * 4, "GBps": g4,
                        "frac": g4 / HBM_PEAK_GBS, "bound": "hbm (write of the masked-out logits; nothing is read)", "mask_bytes_not_written_per_env": A})
            del obs3, logits
        eng.close()
        del ring
        torch.cuda.empty_cache()
    return out


def extras_wrapper(K: int = 40):
    """The tier a trainer consumes: one AttackerVecEnv.step (marlon's AttackerEnvWrapper + MaskedDiscreteAttackerWrapper for the whole batch:
    decode, environment step, observation, bookkeeping, auto-reset) per call, 65 536 Chain-10 envs, valid Discrete actions drawn from the
    action masks.  (a) as round 1 measured it: masks materialised, eager launches; (b) no mask written (the policy masks its logits with
    mcbs_mask_logits), eager launches; (c) the same with the whole step replayed as one hipGraph.  Wall time per call on the host,
    synchronised at both ends."""
    import torch
    from marlon_amd.samples import chainpattern
    from marlon_amd.wrappers import AttackerVecEnv
    E = ENVS_PER_GPU
    kw = dict(maximum_node_count=12, maximum_total_credentials=12, discrete=True)
    ref = AttackerVecEnv(chainpattern.new_environment(10), E, **kw)              # supplies the masks the stand-in policy samples from
    g = torch.Generator(device=ref.engine.device).manual_seed(0)
    acts = []
    for _ in range(K + 5):
        m = ref.action_masks()
        a = torch.where(m, torch.rand(m.shape, generator=g, device=m.device), torch.full((1,), -1.0, device=m.device)).argmax(dim=1)
        acts.append(a)
        ref.step(a)
    ref.close()
    del ref
    torch.cuda.empty_cache()
    out = []
    for name, opts in (("masks materialised, eager launches", dict()), ("no mask materialised, eager launches", dict(materialize_masks=False)),
                       ("no mask materialised, whole step as one hipGraph", dict(materialize_masks=False, use_graph=True))):
        env = AttackerVecEnv(chainpattern.new_environment(10), E, **kw, **opts)
        for a in acts[:5]:
            env.step(a)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for a in acts[5:]:
            _, r, _, _, _ = env.step(a)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        out.append({"workload": f"AttackerVecEnv.step, CyberBattleChain size=10, {E} envs, Discrete actions, {name}", "us_per_step": dt * 1e6,
                    "env_steps_per_s": E / dt, "last_reward_sum": float(r.double().sum())})
        env.close()
        del env
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    sys.exit(main())
