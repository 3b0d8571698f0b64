#!/bin/bash
# r03: full GPU suite + smoke after the torch-fill synchronisation fix
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
echo "== full GPU suite"; timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -25
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
