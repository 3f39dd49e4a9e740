"""Time the UNMODIFIED reference's CyberBattleEnv.step in this container (it cannot travel to the GPU box): BASELINE.json config 1
(Chain-10, tight bounds, attacker only) and the ToyCtf + ScanAndReimage configuration, one process and one process per core.
    python oracle/refharness/time_reference.py            -> JSON lines (kept in profiles/reference_cpu_timing.json)
"step_only" times just env.step(a) with actions sampled beforehand per step; "loop" also counts sample_valid_action and resets."""
from __future__ import annotations

import json
import multiprocessing as mp
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, HERE)


def run(config: str, steps: int, seed: int):
    import logging
    logging.disable(logging.CRITICAL)
    import ref_loader
    ref = ref_loader.load()
    AG = ref.env.AttackerGoal
    if config == "chain10":
        env = ref.CyberBattleChain(size=10, attacker_goal=AG(own_atleast_percent=1.0), maximum_node_count=12, maximum_total_credentials=12,
                                   throws_on_invalid_actions=False)
    else:
        env = ref.CyberBattleToyCtf(attacker_goal=AG(own_atleast=6), defender_agent=ref.defender.ScanAndReimageCompromisedMachines(0.6, 2, 5),
                                    defender_constraint=ref.env.DefenderConstraint(maintain_sla=0.8), maximum_node_count=12,
                                    maximum_total_credentials=10, throws_on_invalid_actions=False)
    env.reset(seed=seed)
    t_step = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        a = env.sample_valid_action()
        s0 = time.perf_counter()
        _, _, done, _, _ = env.step(a)
        t_step += time.perf_counter() - s0
        if done:
            env.reset()
    return steps / t_step, steps / (time.perf_counter() - t0)


def _worker(args):
    return run(*args)


if __name__ == "__main__":
    cores = os.cpu_count() or 1
    for config, steps in (("chain10", 4000), ("toyctf_defender", 4000)):
        so, lo = run(config, steps, 1)
        print(json.dumps(dict(config=config, processes=1, steps=steps, step_only_steps_per_s=round(so), loop_steps_per_s=round(lo))))
        with mp.Pool(cores) as pool:
            t0 = time.perf_counter()
            res = pool.map(_worker, [(config, steps, 10 + i) for i in range(cores)])
            wall = time.perf_counter() - t0
        print(json.dumps(dict(config=config, processes=cores, steps=steps * cores, step_only_steps_per_s_sum=round(sum(r[0] for r in res)),
                              loop_steps_per_s_wall=round(steps * cores / wall))))
