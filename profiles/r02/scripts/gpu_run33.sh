#!/bin/bash
# r02 run 33: workgroups per launch (resident / div) vs burst length
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks']))"; }
for d in 4 5 6 7 8 10; do for args in "--steps 20 --warmup 5" "--steps 200 --warmup 20"; do echo "-- TPT_GRID_DIV=$d $args"; TPT_GRID_DIV=$d timeout 300 python bench.py --no-cpu-baseline --no-extras $args 2>/dev/null | tail -1 | summ; done; done
