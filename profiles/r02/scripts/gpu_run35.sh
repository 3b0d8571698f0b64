#!/bin/bash
# r02 run 35: workgroups per launch for the large workloads in a long stream (C3 4K x16, C5 46 spheres x.. )
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks']))"; }
for f in 200 400 800 1600; do for args in "--workload c3 --steps 60 --warmup 10" "--workload c5 --steps 60 --warmup 10"; do echo "-- TPT_GRID_FILL=$f $args"; TPT_GRID_FILL=$f timeout 300 python bench.py --no-cpu-baseline --no-extras $args 2>/dev/null | tail -1 | summ; done; done
