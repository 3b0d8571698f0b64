#!/bin/bash
# r02 run 23: 64-B path record (colour sum + stack level 0 in LDS): parity, speed, HBM traffic
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms pipe %.4f grid %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['pipeline_ms_per_step'], d['config']['grid_blocks']))"; }
echo "== quick parity"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "per_pixel_bit_exact or config2_1280 or overlap or seventy or animated or sharded or small_scenes or spp" 2>&1 | tail -3
for args in "--steps 20 --warmup 5" "--steps 20 --warmup 5" "--steps 200 --warmup 20" "--workload c3 --steps 20 --warmup 10" "--workload c5 --steps 20 --warmup 10" "--overlap 1 --steps 50" "--overlap 2 --steps 50"; do echo "-- $args"; timeout 300 python bench.py --no-cpu-baseline --no-extras $args 2>&1 | tail -1 | summ; done
echo "== traffic (overlap 1, full grid)"; bash tools/traffic.sh "--no-extras" 2>&1 | tail -6
echo "== traffic (steady-state grid: 64 workgroups)"; TPT_GRID_DIV=8 bash tools/traffic.sh "--no-extras" 2>&1 | tail -6
echo "== stats2"; timeout 60 python tools/stats2_burst.py 2>&1 | grep -v amdgpu.ids
