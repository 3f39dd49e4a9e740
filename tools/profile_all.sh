#!/bin/bash
# rocprofv3 evidence for the numbers bench.py prints (run on the MI355X box from the repo root):
#   tools/profile_all.sh [tag] [workload ...]
# For every workload three separate runs — `--kernel-trace --stats`, `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE` (the two counters
# cannot share a pass on gfx950 and --pmc is never combined with other tracing) — into gpurun_out/prof_<tag>/<workload>/{stats,fetch,write}.
# Summarise afterwards (in the build container, where .git is) with tools/pmc_summary.py workload ... -> profiles/.
set -o pipefail
export TMPDIR=/tmp
tag=${1:-r2}
shift || true
if [ $# -gt 0 ]; then wl=("$@"); else
wl=(bench:headline step:headline step:config3 step:config4 step:config5 obs:headline obs:config3 obs:config4 discrete:headline logits:headline wrapper:headline); fi
root=$(pwd)/gpurun_out/prof_${tag}
mkdir -p "$root"
for w in "${wl[@]}"; do
    d="$root/${w/:/_}"
    mkdir -p "$d"
    k=200; case "$w" in obs:config4) k=5;; obs:*|discrete:*|logits:*) k=20;; esac
    echo "== $w (K=$k)"
    case "$w" in bench:*)   # bench.py ITSELF (headline leg only), the interpreter directly after `--`: every launch of the headline kernel in this
        # process is a graph replay of the timed shape, so the kernel-trace average is the regime bench.py's `kernel_us` describes
        ba=(bench.py --gpus 1 --no-extras --no-cpu-baseline)
        rocprofv3 --kernel-trace --stats --output-format csv -d "$d/stats" -o run -- python3 "${ba[@]}" > "$d/stats.log" 2>&1 || { echo "stats run failed for $w"; tail -5 "$d/stats.log"; exit 1; }
        rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$d/fetch" -o run -- python3 "${ba[@]}" > "$d/fetch.log" 2>&1 || { echo "fetch run failed for $w"; tail -5 "$d/fetch.log"; exit 1; }
        rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$d/write" -o run -- python3 "${ba[@]}" > "$d/write.log" 2>&1 || { echo "write run failed for $w"; tail -5 "$d/write.log"; exit 1; }
        python3 tools/pmc_summary.py workload "$d" || exit 1; continue;; esac
    rocprofv3 --kernel-trace --stats --output-format csv -d "$d/stats" -o run -- python3 tools/profile_run.py "$w" $k > "$d/stats.log" 2>&1 || { echo "stats run failed for $w"; tail -5 "$d/stats.log"; exit 1; }
    case "$w" in wrapper:*) python3 tools/pmc_summary.py workload "$d" || exit 1; continue;; esac     # per-kernel durations inside the graph only
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$d/fetch" -o run -- python3 tools/profile_run.py "$w" $k > "$d/fetch.log" 2>&1 || { echo "fetch run failed for $w"; tail -5 "$d/fetch.log"; exit 1; }
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$d/write" -o run -- python3 tools/profile_run.py "$w" $k > "$d/write.log" 2>&1 || { echo "write run failed for $w"; tail -5 "$d/write.log"; exit 1; }
    case "$w" in step:*)   # instruction mix and where the wavefronts' cycles go (SQ block: 8 counters per pass)
        rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$d/sqi" -o run -- python3 tools/profile_run.py "$w" $k > "$d/sqi.log" 2>&1 || { echo "SQ instruction pass failed for $w"; tail -5 "$d/sqi.log"; }
        rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$d/sqc" -o run -- python3 tools/profile_run.py "$w" $k > "$d/sqc.log" 2>&1 || { echo "SQ cycle pass failed for $w"; tail -5 "$d/sqc.log"; }
        ;; esac
    # summarise on the spot and drop the bulky raw CSVs (gpurun_out is capped at 64 MiB)
    python3 tools/pmc_summary.py workload "$d" || exit 1
done
du -sh "$root"
