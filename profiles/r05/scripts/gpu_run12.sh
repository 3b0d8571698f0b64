#!/bin/bash
# Round 5, GPU call 12: the suite's order around the C5 first-render difference: slots grow, shrink, the hooks build is used and shut
# down (new in round 5) / kept alive (round 4) / not used, then the C5 frame twice.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
for mode in hooks keep none; do echo "== $mode"; timeout 200 python tools/c5_after_hooks.py 6 $mode 2>&1 | grep -v "$F" | tail -8; done
