#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms occ %d grid %d lds %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['blocks_per_cu'], d['config']['grid_blocks'], d['config']['lds_bytes_per_block']))"; }
for lv in 6 4 3 10; do echo "-- lds stack levels $lv"; TPT_LDS_STACK_LEVELS=$lv timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | summ; done
for args in "--fold 1" "--lds-scene 0" "--persistent 2 --fold 0" "--persistent 2 --fold 1" "--workload c3 --steps 10 --warmup 2" "--workload c3 --steps 10 --warmup 2 --persistent 2" "--workload c5 --steps 5 --warmup 1"; do
    echo "-- $args"; timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $args 2>&1 | tail -1 | summ
done
