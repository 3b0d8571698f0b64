#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
for h in 3 0; do echo "== stats burst TPT_HELP=$h"; TPT_HELP=$h timeout 120 python tools/stats_burst.py 2>&1 | grep -v amdgpu.ids; done
