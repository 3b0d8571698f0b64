#!/bin/bash
# Round 6, GPU call 29: the grouped traversal as three dealt stages (super-group entries -> group entries -> survivors; 624 paths per
# workgroup) against the tree before (per-owner loops over super-groups and entry writing); grouped parity; stage profile.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
c5() { for rep in 1 2; do timeout 300 python bench.py --workload c5 --steps 24 --warmup 6 --no-extras --no-cpu-baseline --secondary none 2>&1 | grep -v "$F" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d.get(k) for k in ('value', 'ms_per_step', 'image_fnv')}, d['config']['lds_bytes_per_block'], d['config']['blocks_per_cu'])"; done; }
echo "== C5, working tree (three stages)"; c5
echo "== C5, tree before (af272a5)"; TPT_LIB_DIR=$PWD/tools/_variants/r6_big c5
echo "== C5, working tree again"; c5
echo "== grouped parity"; timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "stress or config5 or group" 2>&1 | grep -v "$F" | tail -6
echo "== stages of the dealing, C5 (stats2 build)"; N=6 timeout 300 python tools/stats2_c5.py 2>&1 | grep -v "$F" | tail -12
