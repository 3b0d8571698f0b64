#!/bin/bash
# Round 5, GPU call 29: the stand-alone probe grown towards the grouped kernel (mode 3: 128 registers, and by feature mask: per-lane gathers
# in flight around the chain, the dealing's LDS protocol, more live registers, the LDS filled to the last byte, scratch).  28 extra streams,
# GPU_MAX_HW_QUEUES=32, 40 launches of ~30 ms per configuration.
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=32
cd "$GRAFT_REPO_ROOT/tools/probes"
P=./mfma_timeslice_probe
for f in 31 0 1 2 3 23; do timeout 60 $P 28 40 3 30 $f; done
