#!/bin/bash
# Round 6, GPU call 19: C5 rate of the one-pass two-level dealing (b91b389) against the tree before it (8a9c282), A/B in one call; the new
# GPU test of deferred sharded frames across state changes; the light profiling build's wave-time sections at C2 (current tree).
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
c5() { for rep in 1 2; do timeout 300 python bench.py --workload c5 --steps 24 --warmup 6 --no-extras --no-cpu-baseline --secondary none 2>&1 | grep -v "$F" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d.get(k) for k in ('value', 'ms_per_step', 'image_fnv', 'parity_ok')}, d['config']['lds_bytes_per_block'], d['config']['blocks_per_cu'])"; done; }
echo "== C5, working tree"; c5
echo "== C5, tree before the one-pass dealing (8a9c282)"; TPT_LIB_DIR=$PWD/tools/_variants/r6_head c5
echo "== C5, working tree again"; c5
echo "== new test"; timeout 600 python -m pytest tests/test_gpu_api.py -x -q -k "deferred or loopback or sharded" 2>&1 | grep -v "$F" | tail -4
echo "== wave-time sections, C2 burst (stats2 build)"; N=40 timeout 300 python tools/stats2_burst.py 2>&1 | grep -v "$F" | tail -14
