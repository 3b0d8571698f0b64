#!/bin/bash
# r04 run 9: grouped traversal with dealt (ray, group) pairs (TPT_GROUP_DEAL) on the 4096-sphere scene: parity, rate, traversal stats
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== grouped-scene parity tests"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -q -x -k "config5 or stress or small_scenes or two_phase_filter_is_conservative or grouped" 2>&1 | grep -v "$F" | tail -6
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d bpc %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks'], d['config']['blocks_per_cu']))"; }
for v in base gd0 base gd0; do
  if [ $v = base ]; then unset TPT_LIB; else export TPT_LIB=$R/tools/_variants/$v/libtoypathtracer_hip.so; fi
  echo "== [$v] c5"; timeout 300 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --workload c5 --steps 20 --warmup 10 2>&1 | tail -1 | summ
done
unset TPT_LIB
echo "== C5 traversal stats (dealt)"; timeout 300 python tools/stats_c5.py 2>&1 | grep -v "$F" | tail -4
echo "== driver's command: burst grids (400 % fill for the first 24 frames, the default) vs stream grids throughout (TPT_GRID_FILL=200)"
for i in 1 2 3; do
  echo "-- default"; timeout 300 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --steps 20 --warmup 5 2>&1 | tail -1 | summ
  echo "-- TPT_GRID_FILL=200"; TPT_GRID_FILL=200 timeout 300 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --steps 20 --warmup 5 2>&1 | tail -1 | summ
done
for n in 30 100; do echo "-- steps $n default / fill 200"; timeout 300 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --steps $n --warmup 5 2>&1 | tail -1 | summ; TPT_GRID_FILL=200 timeout 300 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --steps $n --warmup 5 2>&1 | tail -1 | summ; done
