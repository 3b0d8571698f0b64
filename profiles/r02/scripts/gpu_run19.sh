#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
echo "== stats2 HELP=12"; TPT_HELP=12 timeout 40 python tools/stats2_burst.py 2>&1 | grep -v amdgpu.ids
