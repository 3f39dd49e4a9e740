#!/bin/bash
# SQ instruction / cycle counters of any profile_run workload (two separate rocprofv3 --pmc passes), per kernel:
#   tools/sq_counters.sh kind:workload [K] [kernel-name substring]
set -o pipefail
export TMPDIR=/tmp
w=${1:?kind:workload}; k=${2:-10}; pat=${3:-mcbs::}
d=gpurun_out/sq/${w/:/_}; mkdir -p "$d"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$d/sqi" -o run -- python3 tools/profile_run.py "$w" "$k" > "$d/sqi.log" 2>&1 || { tail -5 "$d/sqi.log"; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$d/sqc" -o run -- python3 tools/profile_run.py "$w" "$k" > "$d/sqc.log" 2>&1 || { tail -5 "$d/sqc.log"; exit 1; }
python3 - "$d" "$pat" <<'PY'
import collections, csv, glob, json, sys
d, pat = sys.argv[1], sys.argv[2]
out = collections.defaultdict(dict)
for sub in ("sqi", "sqc"):
    for f in glob.glob(f"{d}/{sub}/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                per[(r["Kernel_Name"].split("(")[0][:70], r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        agg = collections.defaultdict(list)
        for (kn, _, cn), v in per.items():
            agg[(kn, cn)].append(v)
        for (kn, cn), v in agg.items():
            out[kn][cn] = sum(v) / len(v)
for kn, c in out.items():
    wv, wc = c.get("SQ_WAVES"), c.get("SQ_WAVE_CYCLES")
    row = {"kernel": kn, "waves": wv}
    if wv:
        row.update({n[9:] + "_per_wave": round(v / wv, 1) for n, v in c.items() if n.startswith("SQ_INSTS_")})
    if wc:
        row.update(parked=round(c.get("SQ_WAIT_ANY", 0) / wc, 3), issue_stalled=round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 3), issuing=round(c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3))
    print(json.dumps(row))
PY
rm -rf "$d/sqi" "$d/sqc"
