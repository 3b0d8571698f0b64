#!/bin/bash
# Round 5, GPU call 10: DrawTest of the 4096-sphere frame repeated (the calling pattern of the test that saw 34 extra rays once)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== tree"; timeout 200 python tools/c5_drawtest_determinism.py 40 2>&1 | grep -v "$F" | tail -6
echo "== tail helpers off"; TPT_TAIL_HELPERS=0 timeout 200 python tools/c5_drawtest_determinism.py 20 2>&1 | grep -v "$F" | tail -4
echo "== 120 registers, no uniformHere"; TPT_LIB_DIR=$PWD/tools/_variants/c5v120nouni timeout 200 python tools/c5_drawtest_determinism.py 20 2>&1 | grep -v "$F" | tail -4
