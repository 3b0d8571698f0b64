#!/bin/bash
# Round 5, GPU call 21: with 16 extra streams in the process, which part of the grouped kernel is sensitive?  (1) -DTPT_GROUP_DEAL=0: every
# lane walks its own ray's groups (no pair lists, no parked records, no ds_min_u64); (2) -DTPT_GROUP_DEAL_EXACT=0: no second dealing;
# (3) the shipped kernel on the same 4096 spheres with 8 lights instead of 64; (4) shipped, 64 lights (control).
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
run() { echo "== $1"; n=$2; shift; shift; env "$@" C5_PATH=device C5_DISTURB=torch_streams timeout 400 python tools/c5_after_hooks.py $n keep 2>&1 | grep -v "$F" | grep "results\|rror" | tail -3; }
run "1: no dealing" 40 TPT_LIB_DIR=tools/_variants/nodeal
run "2: no second dealing" 40 TPT_LIB_DIR=tools/_variants/noexact
run "3: shipped, 8 lights" 40 C5_LIGHTS=8
run "4: shipped, 64 lights" 40
