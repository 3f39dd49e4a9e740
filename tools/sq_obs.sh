#!/bin/bash
# SQ counters of the observation kernels, four envs per wavefront (MCBS_QUAD_OBS=1) and a wavefront per env (MCBS_NO_QUAD_OBS=1):  tools/sq_obs.sh <outdir> [workload ...]
set -o pipefail
export TMPDIR=/tmp
out=$1; shift
wl=("$@"); [ ${#wl[@]} -gt 0 ] || wl=(obs:config3 discrete:headline)
mkdir -p "$out"
python3 -c "import sys; sys.path.insert(0, \".\"); from tools import workloads as W; print(\"csrc_sha256\", W.csrc_sha256())"
for w in "${wl[@]}"; do
  for mode in quad wave; do
    if [ $mode = wave ]; then export MCBS_NO_QUAD_OBS=1; unset MCBS_QUAD_OBS; else unset MCBS_NO_QUAD_OBS; export MCBS_QUAD_OBS=1; fi
    d="$out/${w/:/_}_$mode"; mkdir -p "$d"
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$d/sqi" -o run -- python3 tools/profile_run.py "$w" 10 > "$d/sqi.log" 2>&1 || { tail -5 "$d/sqi.log"; exit 1; }
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$d/sqc" -o run -- python3 tools/profile_run.py "$w" 10 > "$d/sqc.log" 2>&1 || { tail -5 "$d/sqc.log"; exit 1; }
    python3 - "$d" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/sq*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "obs_" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    w = sum(c["SQ_WAVES"]) / len(c["SQ_WAVES"]) if c.get("SQ_WAVES") else 0
    print(d.split("/")[-1], k, {n: round(sum(v) / len(v) / (w if n.startswith("SQ_INSTS") and w else 1), 1) for n, v in sorted(c.items())}, flush=True)
PY
    rm -rf "$d/sqi" "$d/sqc"
  done
done
