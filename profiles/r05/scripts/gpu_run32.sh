#!/bin/bash
# Round 5, GPU call 32 (the budget's last seconds): the probe with v_permlane32_swap on the masks and on B operands while gathers are in
# flight (feature 32), alone / with the gathers of feature 1 / with everything.
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=32
cd "$GRAFT_REPO_ROOT/tools/probes"
for f in 32 33 63; do timeout 20 ./mfma_timeslice_probe 28 40 3 30 $f; done
