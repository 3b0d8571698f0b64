#!/bin/bash
# Round 6, GPU call 26: pair-list length against paths per workgroup again, now that one round serves 512 groups (291 pairs per call on average).
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
c5() { for rep in 1 2; do timeout 300 python bench.py --workload c5 --steps 24 --warmup 6 --no-extras --no-cpu-baseline --secondary none 2>&1 | grep -v "$F" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d.get(k) for k in ('value', 'ms_per_step', 'image_fnv')}, d['config']['lds_bytes_per_block'], d['config']['blocks_per_cu'])"; done; }
echo "== shipped (448 entries, 720 paths)"; c5
for v in cap384 cap512 cap576 cap640; do echo "== $v"; TPT_LIB_DIR=$PWD/tools/_variants/r6_$v c5; done
echo "== shipped again"; c5
