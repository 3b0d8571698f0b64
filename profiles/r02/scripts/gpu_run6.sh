#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
echo "== probe default"; timeout 120 python tools/pipeline_probe.py 2>&1 | grep -v amdgpu.ids
echo "== probe TPT_HELP=0"; TPT_HELP=0 timeout 120 python tools/pipeline_probe.py 2>&1 | grep -v amdgpu.ids
