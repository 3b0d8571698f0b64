#!/bin/bash
# Round 6, GPU call 21: call 20's stage profile again with the stage sums kept in LDS (global atomics made the build 60 x slower and the shares meaningless).
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== stages of the dealing, C5 (stats2 build)"; N=6 timeout 300 python tools/stats2_c5.py 2>&1 | grep -v "$F" | tail -24
