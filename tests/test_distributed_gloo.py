"""Multi-GPU path, rehearsed on CPU with gloo (world_size 2): the env batch shards embarrassingly by global env id
(bench.py gives rank r the ids r*E .. r*E+E-1 through mcbs_batch_cfg.env_id_base), the defender's Philox stream is
keyed by the GLOBAL id, so two half-size shards must reproduce one full-size batch bit for bit; the only collectives
are the timing MAX and the all_gather of episode returns.  The stepper here is the CPU oracle (same cfg struct,
same keying) because this container has no GPU; tests/test_gpu_parity.py checks the same property on the engine."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_shard(n_envs, base, steps, actions):
    from marlon_amd import flatten
    from marlon_amd._abi import EnvSpec
    from marlon_amd.samples import toy_ctf
    from oracle.oracle import Oracle
    topo = flatten.flatten(toy_ctf.new_environment())
    spec = EnvSpec(n_envs=n_envs, maximum_node_count=12, maximum_total_credentials=10,
                   attacker_goal=dict(own_atleast=6, own_atleast_percent=1.0), maintain_sla=0.8,
                   defender=("scan_and_reimage", 0.6, 2, 5), auto_reset=True, seed=99, env_id_base=base)
    o = Oracle(topo, spec)
    total = np.zeros(n_envs)
    avail = None
    for t in range(steps):
        out = o.step(actions[t, base:base + n_envs])
        total += out["reward"]
        avail = out["availability"]
    return total, avail


def _worker(rank, world, port, E, steps, actions, q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = E // world
    total, avail = _run_shard(per, rank * per, steps, actions)
    dist.barrier()
    elapsed = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)     # stand-in for the timed region
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    mine = torch.from_numpy(total)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    if rank == 0:
        q.put((torch.cat(gathered).numpy(), float(elapsed.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_shards_equal_one_batch_under_gloo():
    E, steps = 64, 60
    rng = np.random.Generator(np.random.PCG64(5))
    actions = np.zeros((steps, E, 5), np.int32)
    actions[..., 0] = rng.integers(0, 3, (steps, E))
    actions[..., 1] = rng.integers(0, 4, (steps, E))
    actions[..., 2] = rng.integers(0, 4, (steps, E))
    actions[..., 3] = rng.integers(0, 7, (steps, E))
    actions[..., 4] = rng.integers(0, 3, (steps, E))
    full, _ = _run_shard(E, 0, steps, actions)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, E, steps, actions, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered, elapsed = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(gathered, full)        # sharding is invisible in the results
    assert elapsed == 0.002                              # MAX over ranks
    assert full.sum() > 0


def _run_bench(argv, env_extra=None, timeout=240):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


def test_bench_self_launches_ranks_when_invoked_plainly():
    """`python bench.py --gpus 2` with no launcher in the environment (how the driver invokes N = 1) must start the two ranks
    itself and print rank 0's JSON line.  --rehearse keeps it CPU-safe: launcher, backend agreement, barrier / MAX / gathers and
    the JSON assembly run for real, the engine does not (value null)."""
    p, line = _run_bench(["--gpus", "2", "--steps", "7", "--warmup", "2", "--rehearse"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert line is not None and line["n_gpus"] == 2 and line["ranks_in_group"] == 2 and line["steps"] == 7 and line["warmup"] == 2
    assert line["self_launched"] is True and line["collective_backend"] == "gloo" and line["value"] is None
    assert len(line["per_rank_ms"]) == 2 and line["scaling"] == "weak"
    # the 8-GPU configurations' legs run per rank at N > 1 (every rank its shard, MAX over ranks): config 3 is a one-GPU configuration
    assert [c["name"] for c in line["configs"]] == ["config4", "config5"] and all(c["n_gpus"] == 2 for c in line["configs"])
    assert all(c["ms_host_clock_max_over_ranks"] >= 4.0 for c in line["configs"])      # rank 1 sleeps 4 ms: the MAX, not rank 0's 2 ms
    assert line["metric"].startswith("env-steps/sec at 65536 envs, CyberBattleChain-10")


def test_bench_under_an_external_launcher_and_rank_failure_is_an_error():
    """The driver's own multi-GPU form (`python -m torch.distributed.run ... bench.py --gpus N`) does not self-launch again; a
    world size that contradicts --gpus makes every rank exit non-zero, and the launcher reports it."""
    import subprocess
    env = dict(os.environ)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--rehearse", "--steps", "3"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and '"self_launched": false' in lines[0]
    cmd[cmd.index("--gpus") + 1] = "4"                      # 2 ranks started, 4 announced
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode != 0
