#!/bin/bash
# r04 run 13: workgroups per launch for LARGE frames: C3 / C5 / C2 at 200 % and 400 % grid fill (TPT_GRID_FILL), 40 timed frames
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks']))"; }
for wl in c3 c5; do for fill in 200 400 300; do echo "== $wl fill $fill"; TPT_GRID_FILL=$fill timeout 150 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --workload $wl --steps 40 --warmup 20 2>&1 | tail -1 | summ; done; done
for fill in 200 300 400; do echo "== c2 steady fill $fill"; TPT_GRID_FILL=$fill timeout 100 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --steps 200 --warmup 20 2>&1 | tail -1 | summ; done
