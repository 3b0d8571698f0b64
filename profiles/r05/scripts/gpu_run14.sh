#!/bin/bash
# Round 5, GPU call 14: call 13 again for the three variant builds (their directories lacked the hooks library), 100 repetitions each,
# and the tree once more.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
for v in c5v120 c5v128nouni c5v120nouni ""; do
  if [ -n "$v" ]; then export TPT_LIB_DIR=$PWD/tools/_variants/$v; else unset TPT_LIB_DIR; fi
  echo "== [${v:-tree: 128 registers + uniformHere}]"; timeout 300 python tools/c5_after_hooks.py 100 keep 2>&1 | grep -v "$F" | grep "results\|Error\|rror" | tail -4
done
