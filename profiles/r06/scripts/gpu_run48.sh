#!/bin/bash
# Round 6, GPU call 48: members per group (8 since round 4, chosen with the bounds on the matrix cores) once more under the three-stage
# dealing with half-line bounds: 6 / 12 / 16 against 8 at configs[4] (same image hash required).
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
c5() { for rep in 1 2; do timeout 300 python bench.py --workload c5 --steps 24 --warmup 6 --no-extras --no-cpu-baseline --secondary none 2>&1 | grep -v "$F" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d.get(k) for k in ('value', 'ms_per_step', 'image_fnv')}, d['config']['lds_bytes_per_block'], d['config']['blocks_per_cu'], d['config'].get('groups'))"; done; }
echo "== 8 (shipped)"; c5
for G in 6 12 16; do echo "== $G members per group"; TPT_LIB_DIR=$PWD/tools/_variants/r6_g$G c5; done
echo "== 8 again"; c5
