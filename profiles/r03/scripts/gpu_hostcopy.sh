#!/bin/bash
# r03: staged host copies (tptSetHostCopyThreads): host-path tests, rate per thread count
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
echo "== host-path tests"; timeout 300 python -m pytest tests/test_gpu_api.py -m gpu -q --tb=short -k "drawtest or DrawTest or host or trusted or lookahead" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -15
echo "== rate"; timeout 200 python tools/host_drawtest_rate.py 2>&1 | tail -10
