#!/bin/bash
# r02 run 25: queue push with one returning LDS atomic per lane vs the ballot version
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms pipe %.4f grid %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['pipeline_ms_per_step'], d['config']['grid_blocks']))"; }
for i in 1 2; do
for lib in "" "$R/tools/_variants/pushballot/libtoypathtracer_hip.so"; do
for args in "--steps 200 --warmup 20" "--workload c3 --steps 20 --warmup 10"; do echo "-- lib=[$lib] $args"; TPT_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-extras $args 2>&1 | tail -1 | summ; done; done; done
echo "== quick parity"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "per_pixel_bit_exact or config2_1280 or overlap or seventy" 2>&1 | tail -3
