#!/bin/bash
# r02 run 2: stream-depth grid sizing + priming: the driver's command
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms pipe %.4f grid %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['pipeline_ms_per_step'], d['config']['grid_blocks']))"; }
for args in "--steps 20 --warmup 5" "--steps 20 --warmup 5" "--steps 20 --warmup 5 --prime 0" "--steps 20 --warmup 5 --prime 0" "--steps 200 --warmup 20" "--steps 20 --warmup 5 --overlap 8" "--steps 20 --warmup 5 --overlap 12"; do echo "-- $args"; timeout 200 python bench.py --no-cpu-baseline $args 2>&1 | tail -1 | summ; done
for gd in 6 8 12; do echo "-- TPT_GRID_DIV=$gd steps 20"; TPT_GRID_DIV=$gd timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>&1 | tail -1 | summ; done
