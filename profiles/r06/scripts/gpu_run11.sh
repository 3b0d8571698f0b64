#!/bin/bash
# Round 6, GPU call 11: the two-level bounds filter after the refactor (one function for the kernel and its unit test; more groups than
# the LDS area holds: second level from global memory): whole GPU parity file, C5 rates, a 20 000-sphere scene, the time-sliced child.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | grep -v "$F" | tail -8
for hs in 0 3 4; do
  echo "== bench c5 --hit-spheres $hs"
  timeout 300 python bench.py --workload c5 --steps 24 --warmup 6 --no-extras --no-cpu-baseline --secondary none --hit-spheres $hs 2>&1 | grep -v "$F" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d[k] for k in ('value', 'ms_per_step', 'trace_launch_ms_avg', 'image_fnv')}, d['config']['hit_spheres'], d['config']['bounds_on_matrix_cores'], d['config']['lds_bytes_per_block'], d['config']['blocks_per_cu'])"
done
echo "== time-sliced child, 250 sets, default variant"
timeout 900 python tests/c5_timeslice_child.py 250 0 2>&1 | grep -v "$F" | tail -1 | cut -c1-400
