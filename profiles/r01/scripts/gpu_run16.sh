#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 60 python tools/qtest.py 2>&1 | grep -v amdgpu.ids
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms occ %d grid %d lds %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['blocks_per_cu'], d['config']['grid_blocks'], d['config']['lds_bytes_per_block']))"; }
for args in "--overlap 1" "--overlap 2" "--workload c3 --steps 10 --warmup 2 --overlap 1" "--workload c3 --steps 10 --warmup 2"; do echo "-- queue $args"; timeout 60 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --persistent 3 $args 2>&1 | tail -1 | summ; done
for args in "--overlap 1" "--overlap 2" "--workload c3 --steps 10 --warmup 2"; do echo "-- q16p2048 $args"; TPT_LIB=tools/_variants/q16p2048/libtoypathtracer_hip.so timeout 60 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --persistent 3 $args 2>&1 | tail -1 | summ; done
timeout 120 python tools/stats_run.py 2>&1 | grep -v amdgpu.ids | grep -E "==|utilisation|queue|refill"
