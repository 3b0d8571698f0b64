#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms occ %d grid %d lds %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['blocks_per_cu'], d['config']['grid_blocks'], d['config']['lds_bytes_per_block']))"; }
echo "-- v1 recursive LDS stack"; timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | summ
echo "-- v1 recursive global stack"; TPT_GLOBAL_STACK=1 timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | summ
echo "-- v1 recursive global stack c3"; TPT_GLOBAL_STACK=1 timeout 300 python bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | summ
for v in sw2 sw8; do for w in c2 c3; do echo "-- sorted $v $w"; TPT_LIB=tools/_variants/$v/libtoypathtracer_hip.so timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --persistent 2 2>&1 | tail -1 | summ; done; done
