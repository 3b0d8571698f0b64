#!/bin/bash
# Round 6, GPU call 23: the member filter by teams of 8 lanes (one member each, coalesced 128-byte lines; TPT_TEAM_UNROLL steps in flight)
# against the pair-per-lane form (the shipped tree, 4c5...): same-box A/B; image hash of 30 frames must stay a29a64af.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
c5() { for rep in 1 2; do timeout 300 python bench.py --workload c5 --steps 24 --warmup 6 --no-extras --no-cpu-baseline --secondary none 2>&1 | grep -v "$F" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d.get(k) for k in ('value', 'ms_per_step', 'image_fnv')}, d['config']['lds_bytes_per_block'], d['config']['blocks_per_cu'])"; done; }
echo "== C5, shipped (pair per lane)"; c5
for U in 1 2 4; do echo "== teams, unroll $U"; TPT_LIB_DIR=$PWD/tools/_variants/r6_teams$U c5; done
echo "== C5, shipped again"; c5
