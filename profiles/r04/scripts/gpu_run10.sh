#!/bin/bash
# r04 run 10: group bounds on the matrix cores (hitSpheresGroupedDeal + buildGroupMatrixTable): device conservativeness test, the
# grouped-scene parity tests, C5 rate against the VALU-filter / undealt build, traversal stats; C2 numbers of the tree
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== group filter on the device + grouped-scene parity tests"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -q -x -k "group_matrix or config5 or stress or small_scenes or two_phase_filter_is_conservative or grouped" 2>&1 | grep -v "$F" | tail -6
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d bpc %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks'], d['config']['blocks_per_cu']))"; }
for v in base gd0 base; do
  if [ $v = base ]; then unset TPT_LIB; else export TPT_LIB=$R/tools/_variants/$v/libtoypathtracer_hip.so; fi
  echo "== [$v] c5"; timeout 300 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --workload c5 --steps 20 --warmup 10 2>&1 | tail -1 | summ
done
unset TPT_LIB
echo "== [base] c5 with the VALU filter for the bounds (--hit-spheres 3)"; timeout 300 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --workload c5 --hit-spheres 3 --steps 20 --warmup 10 2>&1 | tail -1 | summ
echo "== C5 traversal stats"; timeout 300 python tools/stats_c5.py 2>&1 | grep -v "$F" | tail -4
echo "== C2 driver cmd / steady / c3"
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 2>&1 | tail -1 | summ
timeout 300 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --steps 200 --warmup 20 2>&1 | tail -1 | summ
timeout 300 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --workload c3 --steps 20 --warmup 10 2>&1 | tail -1 | summ
