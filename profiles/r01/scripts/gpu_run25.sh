#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms occ %d grid %d lds %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['blocks_per_cu'], d['config']['grid_blocks'], d['config']['lds_bytes_per_block']))"; }
for ov in 2 3 4; do echo "-- overlap $ov"; timeout 90 python bench.py --no-cpu-baseline --overlap $ov 2>&1 | tail -1 | summ; done
for lv in 5 7 8; do echo "-- lds stack levels $lv"; TPT_LDS_STACK_LEVELS=$lv timeout 90 python bench.py --no-cpu-baseline 2>&1 | tail -1 | summ; done
echo "-- c1 640x360x1"; timeout 90 python bench.py --no-cpu-baseline --workload c1 2>&1 | tail -1 | summ
