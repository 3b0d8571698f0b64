#!/bin/bash
# Round 5, GPU call 24: GPU_MAX_HW_QUEUES=32 and 16 extra streams (the time-sliced condition) -- (A) the dealing kernel WITHOUT the matrix
# cores (bounds through the packed VALU filter, tptSetKernelVariant(3, 3, -1)); (B) one launch in flight and no second grids: a single
# busy queue.
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=32
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
run() { echo "== $1"; n=$2; shift; shift; env "$@" C5_PATH=device C5_DISTURB=torch_streams timeout 400 python tools/c5_after_hooks.py $n keep 2>&1 | grep -v "$F" | grep "results\|rror" | tail -3; }
run "A: dealing, bounds on the VALU" 40 C5_VARIANT=3,3,-1
run "B: one in flight, no second grids" 80 C5_INFLIGHT=1 TPT_TAIL_HELPERS=0
