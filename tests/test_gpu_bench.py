"""The driver's benchmark contract, exercised on the GPU box: `python bench.py --gpus 1 --steps K --warmup W` prints ONE JSON line with the
fields the contract names, the parity flags hold (the metric says "bit-exact vs CPU ref": a run whose checks fail prints value null and exits
non-zero), the roofline block is consistent with the timed region, and the counter files under profiles/ belong to the kernels being timed."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_bench_line_contract_small_run():
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "40", "--warmup", "10", "--cpu-seconds", "1",
                        "--envs-per-gpu", "65536"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    b = json.loads(lines[0])
    assert b["metric"] == "env-steps/sec at 65536 envs, CyberBattleChain-10; bit-exact vs CPU ref" and b["unit"] == "env-steps/s"
    assert b["n_gpus"] == 1 and b["steps"] == 40 and b["warmup"] == 10 and b["higher_is_better"] is True and b["scaling"] == "weak"
    assert b["vs_baseline"] is None and b["dtype"] == "int32" and b["data"] == "synthetic" and "workload" in b["config"]
    assert b["parity_ok"] is True and b["value"] is not None
    assert abs(b["value"] - 65536 * 40 / (b["ms_per_step"] * 40 * 1e-3)) / b["value"] < 1e-6           # value = envs x steps / timed region
    assert b["value"] > 1e9                                                                              # (target: 1e7)
    r = b["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - 348 * 65536 / (r["kernel_us"] * 1e-6) / 1e9) / r["achieved"] < 1e-6       # algorithmic bytes / measured kernel time
    assert r["kernel_us"] <= b["ms_per_step"] * 1e3 * 1.05                                               # device time per launch fits the wall time per step
    assert r["replay_rewards_and_dones_equal_timed_region"] is True and b["step_many"]["rewards_and_dones_equal_timed_region"] is True
    # the layout's own bytes beside the contract's 348 B model, and what binds the launch (latency, from the committed SQ counters)
    assert r["layout_bytes_per_env_step"] == 221 and abs(r["frac_layout"] - r["frac"] * 221 / 348) < 1e-9
    assert r["binding"]["what"].startswith("latency") and r["kernel_us_event_pair_per_launch"] is None
    hr = b["headline_with_resets"]                       # the same kernel with ~1 % of the envs ending (and being re-initialised) per step
    assert hr["episodes_ended"] >= 0.005 * 65536 * 50 and hr["rewards_dones_and_episodes_equal_rehearsal"] is True and hr["us_per_step"] > 0
    src = r["traffic_source"]
    assert src["status"] in ("current", "missing") or src["status"].startswith("stale"), src
    if src["status"] == "current":                                                                       # counters taken on THESE kernel sources
        assert r["traffic"] is not None and 100 * 65536 < r["traffic"] < 400 * 65536
    else:
        assert r["traffic"] is None
    c = b["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["rewards_and_dones_equal_gpu"] is True and c["all_cores"]["rewards_and_dones_equal_gpu"] is True
    assert c["reference_python"]["kind"] == "reference" and c["reference_python"]["value"] > 500
    assert [x["name"] for x in b["configs"]] == ["config3", "config4", "config5"] and all(x["us_per_step"] > 0 for x in b["configs"])
    assert all(x["n_gpus"] == 1 and x["rehearsal_equal"] is True for x in b["configs"])
    assert {x["name"] for x in b["observe"]} >= {"headline", "headline_discrete", "headline_mask_logits", "config3"}
    sw = b["batch_sweep"]                                # fixed part of a step vs the part that grows with the batch
    assert [p["envs"] for p in sw["points"]] == [4096, 16384, 32768, 65536, 131072] and 1.0 < sw["fixed_us"] < 5.0 and sw["ps_per_env"] > 0
    w = b["wrapper"]
    assert len(w) == 3 and w[0]["last_reward_sum"] == w[1]["last_reward_sum"] == w[2]["last_reward_sum"]
    assert w[1]["us_per_step"] < w[0]["us_per_step"] and w[2]["us_per_step"] < w[0]["us_per_step"]
    assert "extras_error" not in b


def test_bench_two_ranks_on_one_gpu():
    """The N > 1 path end to end on the one-GPU box: `python bench.py --gpus 2 --single-device` starts its two ranks itself (both on
    cuda:0), the ranks agree on a collective backend (RCCL refuses two ranks on one device, so gloo), each owns its own 65 536-env shard
    (global env ids rank * E ...), and rank 0 reports the whole-job throughput from the MAX of the per-rank times."""
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--single-device", "--steps", "60", "--warmup", "10",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["ranks_in_group"] == 2 and b["self_launched"] is True and b["collective_backend"] in ("gloo", "nccl")
    assert len(b["per_rank_ms_per_step"]) == 2 and b["parity_ok"] is True and b["scaling"] == "weak"
    assert abs(b["ms_per_step"] - max(b["per_rank_ms_per_step"])) < 1e-9                                # MAX over ranks
    assert abs(b["value"] - 2 * 65536 * 60 / (b["ms_per_step"] * 60 * 1e-3)) / b["value"] < 1e-6       # aggregate over both ranks
    # BASELINE configs 4 and 5 are DEFINED as shards over the GPUs of a node: at N > 1 every rank runs its shard inside the same bracket
    cf = b["configs"]
    assert [x["name"] for x in cf] == ["config4", "config5"] and all(x["n_gpus"] == 2 and x["rehearsal_equal"] is True for x in cf)
    assert [x["envs_per_gpu"] for x in cf] == [8192, 16384]
    for x in cf:                                           # aggregate over both shards, from the MAX-over-ranks host clock
        assert abs(x["env_steps_per_s"] - 2 * x["envs_per_gpu"] / (x["ms_per_step_host_clock_max_over_ranks"] * 1e-3)) / x["env_steps_per_s"] < 1e-6
    assert b["headline_with_resets"]["episodes_ended"] > 0 and "observe" not in b and "extras_error" not in b


def test_build_then_smoke_in_one_process():
    """__graft_entry__.build() loads the library (ABI check) before anything imported torch; smoke() in the same process then runs on the
    HIP runtime the library brought in — the wrong one unless load_library imports torch first ("no ROCm-capable device is detected")."""
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke()"], cwd=REPO, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "smoke ok" in out.stdout, (out.stdout[-1500:], out.stderr[-1500:])
