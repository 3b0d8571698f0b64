#!/bin/bash
# Round 5, GPU call 15: is the rare C5 difference older than this round?  The round-4 library (git c6e2b16) under the same tool; the
# tree with the tail helpers off; the tree without a second library context in the process.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== round-4 library, hooks context kept alive"; TPT_LIB_DIR=$PWD/tools/_variants/r04 timeout 300 python tools/c5_after_hooks.py 100 keep 2>&1 | grep -v "$F" | grep "results\|rror" | tail -3
echo "== tree, helpers off, hooks context kept alive"; TPT_TAIL_HELPERS=0 timeout 300 python tools/c5_after_hooks.py 100 keep 2>&1 | grep -v "$F" | grep "results\|rror" | tail -3
echo "== tree, no second context"; timeout 300 python tools/c5_after_hooks.py 100 none 2>&1 | grep -v "$F" | grep "results\|rror" | tail -3
echo "== tree, hooks context shut down"; timeout 300 python tools/c5_after_hooks.py 100 hooks 2>&1 | grep -v "$F" | grep "results\|rror" | tail -3
