#!/bin/bash
# Round 5, GPU call 27: the MFMA accumulators pinned to v[96:127] (inline-asm MFMAs, -DTPT_MX_EXPERIMENT=4) instead of v[0:31]: a context-save
# handler stores v0-v3 first and the rest of the registers much later -- does a result that is still in the matrix pipeline get lost there?
# GPU_MAX_HW_QUEUES=32 + 16 extra streams.
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=32
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
run() { echo "== $1"; n=$2; shift; shift; env "$@" C5_PATH=device C5_DISTURB=torch_streams timeout 400 python tools/c5_after_hooks.py $n keep 2>&1 | grep -v "$F" | grep "results\|rror" | tail -3 | cut -c1-1500; }
run "4: accumulators in v[96:127]" 60 TPT_LIB_DIR=$PWD/tools/_variants/mxpin
