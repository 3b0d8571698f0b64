#!/bin/bash
# r02 run 48: pool size per workgroup (512 x 4 waves, 1024 x 8 = default, 2048 x 16) with the 64-B record; burst fill 600 / 800 %
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d occ %d lds %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks'], d['config']['blocks_per_cu'], d['config']['lds_bytes_per_block']))"; }
for v in base p2048w16 p512w4; do for args in "--steps 200 --warmup 20" "--steps 20 --warmup 5" "--workload c3 --steps 40 --warmup 10"; do
  lib=""; [ $v != base ] && lib=$GRAFT_REPO_ROOT/tools/_variants/$v/libtoypathtracer_hip.so
  echo "-- $v $args"; TPT_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-extras $args 2>/dev/null | tail -1 | summ; done; done
for f in 600 800; do echo "-- TPT_GRID_FILL=$f --steps 20 --warmup 5"; TPT_GRID_FILL=$f timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 2>/dev/null | tail -1 | summ; done
