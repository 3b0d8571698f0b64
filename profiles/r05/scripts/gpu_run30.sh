#!/bin/bash
# Round 5, GPU call 30 (the last one): the build that differs in 87 % of its renders (-DTPT_MX_EXPERIMENT=2) as the base --
# (a) control at 32 queues + 16 streams; (b) the same build WITHOUT time-slicing (20 queues): does it need the time-slicing at all?
# (c) experiment 5: as 2, and no load ever lands in a register that an MFMA of the same call read as its A operand.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
run() { echo "== $1"; n=$2; shift; shift; env "$@" C5_PATH=device C5_DISTURB=torch_streams timeout 100 python tools/c5_after_hooks.py $n keep 2>&1 | grep -v "$F" | grep "results\|rror" | tail -2 | cut -c1-700; }
run "a: experiment 2, 32 queues + 16 streams" 8 GPU_MAX_HW_QUEUES=32 TPT_LIB_DIR=$PWD/tools/_variants/mxkeep
run "b: experiment 2, 20 queues" 30 GPU_MAX_HW_QUEUES=20 TPT_LIB_DIR=$PWD/tools/_variants/mxkeep
run "c: experiment 5, 32 queues + 16 streams" 20 GPU_MAX_HW_QUEUES=32 TPT_LIB_DIR=$PWD/tools/_variants/mxkeepall
