#!/bin/bash
# launch-shape experiments for the step kernels (run on the MI355X box): every workload under the library's developer switches.
#   (none)                               the defaults: hot image through L1 / L2 with one-wavefront workgroups; G lanes per env beyond 64 nodes
#   MCBS_NO_COOP=1                       one lane per env for every topology (mcbs_step.hip only)
#   MCBS_COOP_MAX_ENVS=<n>               batch size up to which the G-lanes-per-env kernel is chosen (default 32 768)
#   MCBS_LDS_TOPO=1 [MCBS_STEP_BLOCK=b]  hot image staged in LDS per workgroup of b threads (64 / 128 / 256; default: by image size)
# MCBS_STEP_BLOCK without MCBS_LDS_TOPO does nothing (the L1 / L2 variant relies on 64-thread workgroups), so those rows are not swept:
# round 2's "64 vs 256 threads, L1 / L2" columns came from an experiment build that passed the block size as a kernel parameter.
set -o pipefail
wl=${@:-config2 headline config3 config4 config4@65536 config5 config5@131072}
for sw in "" "MCBS_NO_COOP=1" "MCBS_COOP_MAX_ENVS=1000000" "MCBS_LDS_TOPO=1" "MCBS_LDS_TOPO=1 MCBS_STEP_BLOCK=64" "MCBS_LDS_TOPO=1 MCBS_STEP_BLOCK=256"; do
    echo "== switches: ${sw:-none}"
    env $sw python3 tools/bench_configs.py 300 $wl 2>&1 | grep '^{' | cut -c1-112
done
