#!/bin/bash
# r02 run 3: phase 1 on the matrix cores: mask layout test, parity suite, A/B against the VALU filter, SQ counters
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms pipe %.4f grid %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['pipeline_ms_per_step'], d['config']['grid_blocks']))"; }
echo "== new tests"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "matrix or small_scenes" 2>&1 | tail -15
echo "== bench A/B"
for args in "--steps 200 --warmup 20" "--steps 200 --warmup 20 --hit-spheres 3" "--steps 20 --warmup 5" "--steps 20 --warmup 5 --hit-spheres 3" "--workload c3 --steps 20 --warmup 10" "--workload c3 --steps 20 --warmup 10 --hit-spheres 3"; do echo "-- $args"; timeout 300 python bench.py --no-cpu-baseline $args 2>&1 | tail -1 | summ; done
echo "-- vgpr128 variant"; TPT_LIB=$R/tools/_variants/vgpr128/libtoypathtracer_hip.so timeout 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>&1 | tail -1 | summ
echo "-- 32 slots, grid 64, steps 20"; TPT_GRID_DIV=8 timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --overlap 32 2>&1 | tail -1 | summ
echo "-- 24 slots, grid 64, steps 20"; TPT_GRID_DIV=8 timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --overlap 24 2>&1 | tail -1 | summ
echo "== full gpu suite"
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -5
echo "== SQ counters at the steady-state grid (64 workgroups): matrix filter"
TPT_GRID_DIV=8 bash tools/gpu_pmc.sh "--overlap 1" r02m 2>&1 | tail -26
echo "== same, VALU filter"
TPT_GRID_DIV=8 bash tools/gpu_pmc.sh "--overlap 1 --hit-spheres 3" r02v 2>&1 | grep "SQ_INSTS_VALU \|SQ_INSTS_VALU\b\|SQ_BUSY\|GRBM\|SQ_WAVE_CYCLES\|SQ_INSTS_SALU"
echo "== section stats"
timeout 300 python tools/stats_run.py 2>&1 | sed -n 2,40p
