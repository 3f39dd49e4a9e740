#!/usr/bin/env python3
"""Summarise rocprofv3 output directories into the small files kept under profiles/.

  python tools/pmc_summary.py stats  <dir> <out.csv>          kernel-trace --stats summary (top kernels)
  python tools/pmc_summary.py workload <dir>                  <dir> = gpurun_out/prof_<tag>/<kind>_<workload> as tools/profile_all.sh
        leaves it ({stats,fetch,write}/ + stats.log): writes <dir>/summary.json — per mcbs kernel the call count and average /
        minimum duration from `--kernel-trace --stats`, FETCH_SIZE and WRITE_SIZE per launch from the two separate `--pmc`
        passes, HBM-side bytes per launch, and the hash of the kernel sources the counters were taken on — and
        <dir>/kernel_stats.csv (top rows of the stats table), then deletes the bulky raw CSVs.
  python tools/pmc_summary.py collect <prof_root> <profiles_dir> <round_tag>
        copies every <dir>/summary.json to profiles/<round_tag>_<kind>_<workload>.json (adding the git head) and the stats CSVs.

Units and corrections as MI355X_MICROARCH.md's HBM section prescribes: both counters are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of a wide coalesced read, so it is doubled; WRITE_SIZE is taken as is; the two are collected in separate
passes (they cannot share one) and never together with other tracing.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def short(name: str) -> str:
    """mcbs::kernel<template args> without the argument list."""
    n = name.strip('"')
    if n.startswith("void "):
        n = n[5:]
    depth = 0
    for i, ch in enumerate(n):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return n[:i]
    return n


def counter_per_kernel(d, counter):
    fs = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    if not fs:
        return {}
    per_dispatch = collections.defaultdict(float)
    kern = {}
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] == counter:
            per_dispatch[r["Dispatch_Id"]] += float(r["Counter_Value"])
            kern[r["Dispatch_Id"]] = short(r["Kernel_Name"])
    agg = collections.defaultdict(list)
    for k, v in per_dispatch.items():
        agg[kern[k]].append(v)
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def trace_percentiles(d):
    """Per kernel: (median, p10, p90) of the dispatch durations in the kernel trace, in us — the average of a --stats table is pulled up by
    the few launches the profiler's own time-stamping delays."""
    fs = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
    if not fs:
        return {}
    durs = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        try:
            durs[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        except (KeyError, ValueError):
            continue
    out = {}
    for k, v in durs.items():
        v.sort()
        out[k] = (v[len(v) // 2], v[len(v) // 10], v[(len(v) * 9) // 10])
    return out


def stats_rows(d):
    fs = glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True)
    return list(csv.reader(open(fs[0]))) if fs else []


def main():
    mode = sys.argv[1]
    if mode == "stats":
        rows = stats_rows(sys.argv[2])
        with open(sys.argv[3], "w", newline="") as o:
            w = csv.writer(o)
            for r in rows[:8]:
                r[0] = r[0][:120]
                w.writerow(r)
        print(open(sys.argv[3]).read())
    elif mode == "workload":
        from tools import workloads as W
        d = sys.argv[2].rstrip("/")
        rows = stats_rows(f"{d}/stats")
        pct = trace_percentiles(f"{d}/stats")
        fetch = counter_per_kernel(f"{d}/fetch", "FETCH_SIZE")
        write = counter_per_kernel(f"{d}/write", "WRITE_SIZE")
        run = {}
        try:
            run = json.loads([l for l in open(f"{d}/stats.log") if l.startswith("{")][-1])
        except Exception:
            pass
        kernels = []
        for r in rows[1:]:
            name = short(r[0])
            if not name.startswith("mcbs::"):
                continue
            f, nf = fetch.get(name, (None, 0))
            w, nw = write.get(name, (None, 0))
            k = dict(kernel=name, calls=int(r[1]), avg_us=float(r[3]) / 1e3, min_us=float(r[5]) / 1e3, max_us=float(r[6]) / 1e3,
                     share_of_gpu_time_pct=float(r[4]), FETCH_SIZE_KiB_per_launch=f, WRITE_SIZE_KiB_per_launch=w,
                     pmc_dispatches=[nf, nw])
            if name in pct:
                k["median_us"], k["p10_us"], k["p90_us"] = pct[name]
            sq = {}
            for sub, names in (("sqi", ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")),
                               ("sqc", ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"))):
                for cn in names:
                    val = counter_per_kernel(f"{d}/{sub}", cn).get(name)
                    if val is not None:
                        sq[cn] = val[0]
            if sq.get("SQ_WAVES"):
                wv = sq["SQ_WAVES"]
                k["per_wave_instructions"] = {c[9:]: round(sq[c] / wv, 1) for c in sq if c.startswith("SQ_INSTS_")}
                k["waves_per_launch"] = wv
            if sq.get("SQ_WAVE_CYCLES"):
                wc = sq["SQ_WAVE_CYCLES"]
                k["wave_cycle_split"] = {"wave_cycles_per_launch": wc, "parked_at_waitcnt_or_barrier": round(sq.get("SQ_WAIT_ANY", 0) / wc, 3),
                                         "issue_stalled": round(sq.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                                         "issuing": round(sq.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)}
            if f is not None and w is not None:
                k["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0
                envs = run.get("envs") or (run.get("config") or {}).get("envs_per_gpu")
                if envs:
                    k["hbm_bytes_per_env"] = k["hbm_bytes_per_launch"] / envs
                k["hbm_GBps_at_avg"] = k["hbm_bytes_per_launch"] / (k["avg_us"] * 1e-6) / 1e9
            kernels.append(k)
        prog = (f"tools/profile_run.py {run.get('what', '?')}:{run.get('workload', '?')} {run.get('launches', '')}" if "what" in run else
                f"bench.py --gpus 1 --steps {run.get('steps')} --warmup {run.get('warmup')} --no-extras --no-cpu-baseline")
        if "metric" in run:      # bench.py's own line: keep what the profile is to be compared with, drop the rest
            run = {k: run[k] for k in ("metric", "value", "steps", "warmup", "ms_per_step", "config", "roofline") if k in run}
        out = dict(run=run, command=f"rocprofv3 {{--kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE}} -- python3 {prog}",
                   csrc_sha256=W.csrc_sha256(), kernels=kernels,
                   note="hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB: FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies the 128-byte "
                        "requests of wide coalesced reads at 64 bytes); counters are L2 memory-side requests, Infinity-Cache hits included")
        json.dump(out, open(f"{d}/summary.json", "w"), indent=1)
        with open(f"{d}/kernel_stats.csv", "w", newline="") as o:
            w = csv.writer(o)
            for r in rows[:10]:
                r[0] = r[0][:140]
                w.writerow(r)
        for sub in ("stats", "fetch", "write", "sqi", "sqc"):
            shutil.rmtree(f"{d}/{sub}", ignore_errors=True)
        for k in kernels[:4]:
            print(json.dumps({a: k.get(a) for a in ("kernel", "calls", "avg_us", "min_us", "hbm_bytes_per_env", "hbm_GBps_at_avg")}))
    elif mode == "collect":
        from tools import workloads as W
        root, dst, tag = sys.argv[2], sys.argv[3], sys.argv[4]
        for d in sorted(glob.glob(f"{root}/*/")):
            name = os.path.basename(d.rstrip("/"))
            if not os.path.exists(f"{d}/summary.json"):
                continue
            s = json.load(open(f"{d}/summary.json"))
            s["git_head_when_collected"] = W.git_head()
            s["csrc_matches_tree_when_collected"] = s.get("csrc_sha256") == W.csrc_sha256()
            json.dump(s, open(f"{dst}/{tag}_{name}.json", "w"), indent=1)
            shutil.copy(f"{d}/kernel_stats.csv", f"{dst}/{tag}_{name}_kernel_stats.csv")
            print(f"{dst}/{tag}_{name}.json")
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
