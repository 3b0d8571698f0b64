#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms occ %d grid %d lds %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['blocks_per_cu'], d['config']['grid_blocks'], d['config']['lds_bytes_per_block']))"; }
for v in sw8 sw16; do for w in "c2" "c3 --steps 10 --warmup 2"; do for ov in 1 2; do echo "-- sorted $v $w overlap $ov"; TPT_LIB=tools/_variants/$v/libtoypathtracer_hip.so timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --persistent 2 --overlap $ov --workload $w 2>&1 | tail -1 | summ; done; done; done
