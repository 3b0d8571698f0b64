#!/bin/bash
# r03: host-pointer DrawTest: look-ahead depth x grid fill x copy threads; the look-ahead tests at the new depths
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
echo "== look-ahead tests"; timeout 300 python -m pytest tests/test_gpu_api.py -m gpu -q --tb=short -x -k "lookahead" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -15
echo "== rate"; timeout 150 python tools/host_drawtest_rate.py 2>&1 | grep -v amdgpu.ids | tail -40
