#!/bin/bash
# Round 5, GPU call 17: where does the rare C5 difference live?  (a) DrawTest without look-ahead (one launch in flight), (b) tptDrawDevice,
# three frames in flight, no host path, (c) the same with one frame in flight, (d) tail helpers off with (b).  Hooks context used and kept.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
run() { echo "== $1"; shift; env "$@" timeout 300 python tools/c5_after_hooks.py 60 keep 2>&1 | grep -v "$F" | grep "results\|rror" | tail -3; }
run "DrawTest, no look-ahead" C5_LOOKAHEAD=0
run "tptDrawDevice, 3 in flight" C5_PATH=device C5_INFLIGHT=3
run "tptDrawDevice, 1 in flight" C5_PATH=device C5_INFLIGHT=1
run "tptDrawDevice, 3 in flight, helpers off" C5_PATH=device C5_INFLIGHT=3 TPT_TAIL_HELPERS=0
