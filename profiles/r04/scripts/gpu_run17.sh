#!/bin/bash
# r04 run 17: 8 members per group as the default: grouped-scene parity tests, C5 rate, traversal stats
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -q -x --timeout=150 -k "group_matrix or config5 or custom_scene or two_phase_filter_is_conservative or both_kernels_full_size or hit_spheres_kernel or small_scenes" 2>&1 | grep -v "$F" | tail -4
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks']))"; }
for i in 1 2; do echo "== c5"; timeout 120 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --workload c5 --steps 40 --warmup 20 2>&1 | tail -1 | summ; done
echo "== c5, 20 frames"; timeout 120 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --workload c5 --steps 20 --warmup 10 2>&1 | tail -1 | summ
echo "== C5 traversal stats"; timeout 90 python tools/stats_c5.py 2>&1 | grep -v "$F" | tail -4
