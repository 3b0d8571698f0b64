#!/bin/bash
# r03: host-pointer DrawTest served from look-ahead batches (per-pixel seeds): host-path tests, rate with / without
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
echo "== host-path tests"; timeout 300 python -m pytest tests/test_gpu_api.py -m gpu -q --tb=short -x -k "drawtest or DrawTest or host or trusted or lookahead" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -15
echo "== rate, batches"; timeout 100 python tools/host_drawtest_rate.py 2>&1 | tail -7
echo "== rate, one launch per frame"; TPT_HOST_BATCH=0 timeout 100 python tools/host_drawtest_rate.py 2>&1 | tail -7
