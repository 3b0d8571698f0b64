#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
echo "== pytest gpu"; timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms occ %d grid %d lds %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['blocks_per_cu'], d['config']['grid_blocks'], d['config']['lds_bytes_per_block']))"; }
for co in 1 0; do for args in "" "--overlap 1" "--fold 1" "--workload c3 --steps 10 --warmup 6" "--workload c5 --steps 6 --warmup 5"; do echo "-- cost_order=$co $args"; TPT_COST_ORDER=$co timeout 90 python bench.py --steps 200 --warmup 20 --no-cpu-baseline $args 2>&1 | tail -1 | summ; done; done
for m in 10 12 14; do echo "-- cost_order=1 maxblocks $m"; TPT_MAX_BLOCKS_PER_CU=$m timeout 90 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | summ; done
