#!/bin/bash
# r02 run 43: why did --animate drop to 31 Gray/s?  pacing on/off x slot factor
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks']))"; }
for env in "TPT_HOST_PACE=0 TPT_SLOT_FACTOR=1" "TPT_HOST_PACE=0 TPT_SLOT_FACTOR=2" "TPT_HOST_PACE=1 TPT_SLOT_FACTOR=2"; do echo "-- $env --animate"; env $env timeout 300 python bench.py --no-cpu-baseline --no-extras --animate 2>/dev/null | tail -1 | summ; done
