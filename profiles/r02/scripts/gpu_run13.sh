#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
for h in 3 1 0; do echo "== stats2 burst TPT_HELP=$h"; TPT_HELP=$h timeout 40 python tools/stats2_burst.py 2>&1 | grep -v amdgpu.ids; done
