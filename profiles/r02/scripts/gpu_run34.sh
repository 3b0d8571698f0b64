#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d host %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks'], d.get('drawtest_host_ms')))"; }
for args in "--steps 20 --warmup 5" "--steps 20 --warmup 5" "--steps 200 --warmup 20" "--steps 50 --warmup 5" "--workload c3 --steps 20 --warmup 10" "--animate"; do echo "-- $args"; timeout 300 python bench.py --no-cpu-baseline $args 2>/dev/null | tail -1 | summ; done
