#!/bin/bash
# Round 5, GPU call 28: the stand-alone probe (tools/probes/mfma_timeslice_probe.hip, no library code): the MFMA chain of matrixApply on A
# tiles from global memory / staged in LDS / the same loads without MFMAs, 40 launches of ~30 ms each, with and without 28 extra streams
# at GPU_MAX_HW_QUEUES=32.
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=32
cd "$GRAFT_REPO_ROOT/tools/probes"
P=./mfma_timeslice_probe
timeout 60 $P 28 40 0
timeout 60 $P 0 40 0
timeout 60 $P 28 40 1
timeout 60 $P 28 40 2
timeout 60 $P 28 40 0 8
