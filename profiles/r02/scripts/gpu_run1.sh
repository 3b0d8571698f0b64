#!/bin/bash
# r02 run 1: baseline after the slot pre-allocation fix: driver's command, short runs, section stats.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms pipe %.4f grid %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['pipeline_ms_per_step'], d['config']['grid_blocks']))"; }
echo "== driver command x3"
for i in 1 2 3; do timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | summ; done
echo "== short/long sweeps"
for args in "--steps 20 --warmup 20" "--steps 50 --warmup 5" "--steps 200 --warmup 20" "--steps 20 --warmup 5 --overlap 8" "--steps 20 --warmup 5 --overlap 4"; do echo "-- $args"; timeout 200 python bench.py --no-cpu-baseline $args 2>&1 | tail -1 | summ; done
echo "== gpu tests (quick subset)"
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
echo "== section stats"
timeout 300 python tools/stats_run.py 2>&1 | head -60
