#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms occ %d grid %d lds %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['blocks_per_cu'], d['config']['grid_blocks'], d['config']['lds_bytes_per_block']))"; }
run() { echo "-- $1 | lv=$2 | $3"; TPT_LDS_STACK_LEVELS=$2 TPT_LIB=$1 timeout 90 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $3 2>&1 | tail -1 | summ; }
D=toypathtracer_amd/lib/libtoypathtracer_hip.so
run $D 6 ""
run tools/_variants/p2pf/libtoypathtracer_hip.so 6 ""
run tools/_variants/w5/libtoypathtracer_hip.so 4 ""
run tools/_variants/w5p2pf/libtoypathtracer_hip.so 4 ""
run tools/_variants/w6/libtoypathtracer_hip.so 3 ""
run $D 6 "--fold 1"
run tools/_variants/w5/libtoypathtracer_hip.so 4 "--fold 1"
run tools/_variants/w5p2pf/libtoypathtracer_hip.so 4 "--fold 1"
run $D 6 "--workload c3 --steps 10 --warmup 2"
run tools/_variants/w5/libtoypathtracer_hip.so 4 "--workload c3 --steps 10 --warmup 2"
run tools/_variants/w5p2pf/libtoypathtracer_hip.so 4 "--workload c3 --steps 10 --warmup 2"
run tools/_variants/p2pf/libtoypathtracer_hip.so 6 "--workload c3 --steps 10 --warmup 2"
run $D 6 "--workload c5 --steps 5 --warmup 1"
run tools/_variants/w5/libtoypathtracer_hip.so 4 "--workload c5 --steps 5 --warmup 1"
