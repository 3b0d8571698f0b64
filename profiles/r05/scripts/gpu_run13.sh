#!/bin/bash
# Round 5, GPU call 13: which change makes the C5 frame through DrawTest come back with a different pixel once in ~40 renders?
# 40 repetitions (80 renders) of tools/c5_after_hooks.py, hooks context kept alive, on four builds.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
for v in "" c5v120 c5v128nouni c5v120nouni; do
  if [ -n "$v" ]; then export TPT_LIB_DIR=$PWD/tools/_variants/$v; else unset TPT_LIB_DIR; fi
  echo "== [${v:-tree: 128 registers + uniformHere}]"; timeout 240 python tools/c5_after_hooks.py 40 keep 2>&1 | grep -v "$F" | grep "results\|f6adf9ed\|rep [0-9]*: \[(1304197[0-9]*, '[0-9a-f]*'), (1304197[0-9]*, '[0-9a-f]*')\]" | grep -v "54197be7'), (130419700, '54197be7" | tail -8
done
