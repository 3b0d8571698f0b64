#!/bin/bash
# Round 5, GPU call 11: the first render after a scene change (one wrong pixel in 2 of 4 suite runs): repeated, on the tree and on
# the build closest to round 4's kernels (120 registers, no de-hoisting).
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== tree"; timeout 300 python tools/c5_first_render.py 24 2>&1 | grep -v "$F" | tail -12
echo "== tree, tail helpers off"; TPT_TAIL_HELPERS=0 timeout 300 python tools/c5_first_render.py 12 2>&1 | grep -v "$F" | tail -8
echo "== 120 registers, no uniformHere"; TPT_LIB_DIR=$PWD/tools/_variants/c5v120nouni timeout 300 python tools/c5_first_render.py 24 2>&1 | grep -v "$F" | tail -12
