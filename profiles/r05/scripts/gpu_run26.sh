#!/bin/bash
# Round 5, GPU call 26: is it the software-managed distance between an MFMA and the first read of its result?  The shipped code with 32
# more idle wait states behind every MFMA chain (-DTPT_MX_EXPERIMENT=3), GPU_MAX_HW_QUEUES=32 + 16 extra streams.
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=32
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
run() { echo "== $1"; n=$2; shift; shift; env "$@" C5_PATH=device C5_DISTURB=torch_streams timeout 400 python tools/c5_after_hooks.py $n keep 2>&1 | grep -v "$F" | grep "results\|rror" | tail -3 | cut -c1-1500; }
run "3: 32 more wait states behind the MFMAs" 50 TPT_LIB_DIR=$PWD/tools/_variants/mxnop
