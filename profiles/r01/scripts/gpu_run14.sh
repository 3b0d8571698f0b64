#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 120 python tools/stats_run.py 2>&1 | grep -v amdgpu.ids
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms occ %d grid %d lds %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['blocks_per_cu'], d['config']['grid_blocks'], d['config']['lds_bytes_per_block']))"; }
for v in q8 q16p512 q8p1024; do for args in "--overlap 1" "--overlap 2" "--workload c3 --steps 10 --warmup 2"; do echo "-- $v $args"; TPT_LIB=tools/_variants/$v/libtoypathtracer_hip.so timeout 60 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --persistent 3 $args 2>&1 | tail -1 | summ; done; done
