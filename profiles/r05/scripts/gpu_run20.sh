#!/bin/bash
# Round 5, GPU call 20: with 16 extra streams in the process (the trigger found in call 19) -- (X) the spill-free build of the grouped
# kernel (tools/_variants/g96w2: 166 registers, no scratch), (Y) the kernel that stages the scene in LDS on launches as long as the
# 4096-sphere ones (default scene, 3840x2160, 64 spp).
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
run() { echo "== $1"; n=$2; shift; shift; env "$@" C5_PATH=device C5_DISTURB=torch_streams timeout 400 python tools/c5_after_hooks.py $n keep 2>&1 | grep -v "$F" | grep "results\|rror" | tail -3; }
run "X: spill-free grouped kernel, 4096 spheres" 40 TPT_LIB_DIR=tools/_variants/g96w2
run "Y: default scene 3840x2160 at 64 spp" 20 C5_SCENE=default4k
