#!/bin/bash
# What does the host's share of a SHORT timed region (the driver's `--steps 20 --warmup 5`: 0.1 ms of device work) respond to?
# Runs bench.py's headline leg under a few HIP-runtime settings and prints value / ms_per_step / kernel_us and the host breakdown.
#   tools/short_run_env.sh [steps] [warmup]        (on the MI355X box, from the repo root)
steps=${1:-20}; warm=${2:-5}
run() {
    label=$1; shift
    for i in 1 2 3; do
        env "$@" python3 bench.py --gpus 1 --steps "$steps" --warmup "$warm" --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=b['config']['timed_region_breakdown_us_rank0']
print('$label', 'value %.4g' % b['value'], 'us/step %.3f' % (b['ms_per_step']*1e3), 'kernel_us %.3f' % b['roofline']['kernel_us'], {k: round(v,1) for k,v in h.items()})"
    done
}
run default X=1
run active_wait_1ms ROC_ACTIVE_WAIT_TIMEOUT=1000
run active_wait_cpu ROC_ACTIVE_WAIT_TIMEOUT=1000 ROC_CPU_WAIT_FOR_SIGNAL=1
run no_graph_pkt_capture DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run batch64 DEBUG_HIP_GRAPH_BATCH_SIZE=64
