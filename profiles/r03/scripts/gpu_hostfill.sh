#!/bin/bash
# r03: host-pointer DrawTest: do blends find free slots when the trace launches leave part of the machine empty?
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
FILLS=85,70,55,40 AHEADS=2,3,4 timeout 150 python tools/host_drawtest_rate.py 2>&1 | grep -v amdgpu.ids | grep -v "copy threads [128]" | tail -40
