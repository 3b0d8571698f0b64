#!/bin/bash
# Round 5, GPU call 31: call 30's (a) and (c) again -- there the library's new rule (more than 22 queues -> groups' bounds on the VALU) had
# switched the matrix-core filter off.  HIP is started with 32 queues first, then the library sees 20 (C5_BYPASS_GUARD).
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
run() { echo "== $1"; n=$2; shift; shift; env "$@" C5_BYPASS_GUARD=1 C5_PATH=device C5_DISTURB=torch_streams timeout 60 python tools/c5_after_hooks.py $n keep 2>&1 | grep -v "$F" | grep "results\|rror" | tail -2 | cut -c1-500; }
run "a: experiment 2, 32 queues + 16 streams" 8 GPU_MAX_HW_QUEUES=32 TPT_LIB_DIR=$PWD/tools/_variants/mxkeep
run "c: experiment 5, 32 queues + 16 streams" 16 GPU_MAX_HW_QUEUES=32 TPT_LIB_DIR=$PWD/tools/_variants/mxkeepall
