#!/bin/bash
# Round 6, GPU call 10: the new default for grouped scenes (two-level packed VALU filter, the groups' pair records in LDS, 816 paths per
# workgroup): parity tests, the time-sliced child process, and the rates of the three variants at configs[4].
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "config5 or time_sliced or stress or sphere_count or small_scenes or custom_scene or both_kernels or hit_spheres_kernel or group_matrix" 2>&1 | grep -v "$F" | tail -15
for hs in 0 3 4; do
  echo "== bench c5 --hit-spheres $hs"
  timeout 300 python bench.py --workload c5 --steps 24 --warmup 6 --no-extras --no-cpu-baseline --secondary none --hit-spheres $hs 2>&1 | grep -v "$F" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d[k] for k in ('value', 'ms_per_step', 'trace_launch_ms_avg', 'image_fnv')}, d['config']['hit_spheres'], d['config']['bounds_on_matrix_cores'], d['config']['lds_bytes_per_block'], d['config']['blocks_per_cu'])"
done
echo "== time-sliced child, 100 sets, default variant / flat / matrix cores"
for v in 0 3 4; do timeout 600 python tests/c5_timeslice_child.py 100 $v 2>&1 | grep -v "$F" | tail -1 | cut -c1-400; done
