#!/bin/bash
# launch-shape experiments for the step kernel (run on the MI355X box): every workload under the library's developer switches
set -o pipefail
wl=${@:-config2 headline config3 config4 config4@65536 config5 config5@131072}
for sw in "" "MCBS_LDS_TOPO=1" "MCBS_LDS_TOPO=1 MCBS_STEP_BLOCK=64" "MCBS_LDS_TOPO=1 MCBS_STEP_BLOCK=256" "MCBS_STEP_BLOCK=64" "MCBS_STEP_BLOCK=128" "MCBS_STEP_BLOCK=256"; do
    echo "== switches: ${sw:-none}"
    env $sw python3 tools/bench_configs.py 300 $wl 2>&1 | grep '^{' | cut -c1-112
done
