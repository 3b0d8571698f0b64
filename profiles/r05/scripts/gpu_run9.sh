#!/bin/bash
# Round 5, GPU call 9: the 4096-sphere scene rendered twice gave 130419700 and 130419734 rays (call 8, tests/test_gpu_parity.py:559).
# Which change?  The same frame over and over on: the tree (128 registers + uniformHere), without uniformHere, 120 registers, both off.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
for v in "" nouni c5v120 c5v120nouni; do
  if [ -n "$v" ]; then export TPT_LIB_DIR=$PWD/tools/_variants/$v; else unset TPT_LIB_DIR; fi
  echo "== [${v:-tree: 128 registers, uniformHere}]"
  timeout 120 python tools/c5_determinism.py c5 10 2>&1 | grep -v "$F" | tail -11
done
unset TPT_LIB_DIR
echo "== c2 on the tree"; timeout 120 python tools/c5_determinism.py c2 8 2>&1 | grep -v "$F" | tail -9
