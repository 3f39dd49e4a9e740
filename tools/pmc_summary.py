#!/usr/bin/env python3
"""Summarise rocprofv3 output directories into the small files kept under profiles/.

  python tools/pmc_summary.py stats  <dir> <out.csv>          kernel-trace --stats summary (top kernels)
  python tools/pmc_summary.py traffic <fetch_dir> <write_dir> <out.json> [envs]
        HBM traffic per launch of mcbs::step_kernel from two separate --pmc passes (FETCH_SIZE, WRITE_SIZE cannot
        share a pass on gfx950: MI355X_MICROARCH.md "rocprofv3 PMC slots").  Units and corrections as that guide's
        HBM section prescribes: both counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide
        coalesced read, so it is doubled; WRITE_SIZE is taken as is.
"""
import collections
import csv
import glob
import json
import sys


def counter_avg(d, counter, kernel_substr):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    per_dispatch = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter:
            per_dispatch[r["Dispatch_Id"]] += float(r["Counter_Value"])
    vals = list(per_dispatch.values())
    return sum(vals) / max(1, len(vals)), len(vals)


if sys.argv[1] == "stats":
    f = glob.glob(f"{sys.argv[2]}/**/*kernel_stats.csv", recursive=True)[0]
    rows = list(csv.reader(open(f)))
    with open(sys.argv[3], "w", newline="") as o:
        w = csv.writer(o)
        for r in rows[:8]:
            r[0] = r[0][:120]
            w.writerow(r)
    print(open(sys.argv[3]).read())
else:
    fetch, nf = counter_avg(sys.argv[2], "FETCH_SIZE", "step_kernel")
    write, nw = counter_avg(sys.argv[3], "WRITE_SIZE", "step_kernel")
    envs = int(sys.argv[5]) if len(sys.argv) > 5 else 65536
    out = dict(kernel="mcbs::step_kernel<0>", envs_per_launch=envs, dispatches=[nf, nw],
               FETCH_SIZE_KiB_per_launch=fetch, WRITE_SIZE_KiB_per_launch=write,
               bytes_per_launch=(2.0 * fetch + write) * 1024.0,
               bytes_per_env_step=(2.0 * fetch + write) * 1024.0 / envs,
               note="FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read); the "
                    "batch state (~35 MB at 65536 Chain-10 envs) is resident in L2 / Infinity Cache between launches, "
                    "so fabric-side traffic is below the algorithmic byte count")
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    print(json.dumps(out, indent=1))
