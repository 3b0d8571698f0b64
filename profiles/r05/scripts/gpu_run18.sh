#!/bin/bash
# Round 5, GPU call 18: WHAT about the second context makes the 4096-sphere frame differ now and then?  tptDrawDevice, three frames
# in flight, second context kept; what it does between the renders varies.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
run() { echo "== $1"; shift; env "$@" C5_PATH=device timeout 300 python tools/c5_after_hooks.py 40 keep 2>&1 | grep -v "$F" | grep "results\|rror\|pixel" | tail -12; }
run "second context: initialised only" C5_DISTURB=hooks_init
run "second context: 20 frames on the shipped kernel" C5_DISTURB=hooks_queue
run "second context: 20 frames on the lane-refill kernel" C5_DISTURB=hooks_lane
run "the same, then 0.3 s idle" C5_DISTURB=hooks_lane_sleep
run "no second context; torch matrix products" C5_DISTURB=torch
