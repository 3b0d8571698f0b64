#!/bin/bash
# r04 run 15: C5 with the big spheres filtered before their exact tests; group size 8 / 16 / 32; grouped-scene parity tests
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -q -x --timeout=150 -k "group_matrix or config5 or custom_scene or two_phase_filter_is_conservative or both_kernels_full_size" 2>&1 | grep -v "$F" | tail -4
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks']))"; }
for v in base g8 g32 base; do
  if [ $v = base ]; then unset TPT_LIB; else export TPT_LIB=$R/tools/_variants/$v/libtoypathtracer_hip.so; fi
  echo "== [$v] c5"; timeout 120 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --workload c5 --steps 40 --warmup 20 2>&1 | tail -1 | summ
done
