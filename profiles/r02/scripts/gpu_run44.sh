#!/bin/bash
# r02 run 44: full GPU suite with paced host + two colour slots per stream + scene ring grown at once; animate bench
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks']))"; }
for args in "--animate" "--animate --steps 20 --warmup 5" "--steps 20 --warmup 5"; do echo "-- $args"; timeout 300 python bench.py --no-cpu-baseline --no-extras $args 2>/dev/null | tail -1 | summ; done
timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -5
