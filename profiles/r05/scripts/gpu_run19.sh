#!/bin/bash
# Round 5, GPU call 19: is the trigger of the rare 4096-sphere difference the NUMBER OF HARDWARE QUEUES the process holds (more queues than the
# device has slots for -> the scheduler time-slices them -> running waves are context-switched)?  tptDrawDevice, three frames in flight.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
run() { echo "== $1"; shift; env "$@" C5_PATH=device timeout 300 python tools/c5_after_hooks.py 40 keep 2>&1 | grep -v "$F" | grep "results\|rror" | tail -3; }
run "no second context; 16 extra torch streams, GPU_MAX_HW_QUEUES=32" C5_DISTURB=torch_streams
run "no second context; 6 extra torch streams" C5_DISTURB=torch_streams C5_TORCH_STREAMS=6
run "second context initialised; GPU_MAX_HW_QUEUES=20" C5_DISTURB=hooks_init GPU_MAX_HW_QUEUES=20
run "second context initialised; GPU_MAX_HW_QUEUES=24" C5_DISTURB=hooks_init GPU_MAX_HW_QUEUES=24
run "second context initialised; GPU_MAX_HW_QUEUES=28" C5_DISTURB=hooks_init GPU_MAX_HW_QUEUES=28
