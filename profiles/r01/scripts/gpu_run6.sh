#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms occ %d grid %d lds %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['blocks_per_cu'], d['config']['grid_blocks'], d['config']['lds_bytes_per_block']))"; }
for p in 1 2; do for fold in 0 1; do for ov in 1 2; do echo "-- persist $p fold $fold overlap $ov"; timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --persistent $p --fold $fold --overlap $ov 2>&1 | tail -1 | summ; done; done; done
echo "-- c3 sorted"; timeout 300 python bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline --persistent 2 2>&1 | tail -1 | summ
echo "-- c5 sorted"; timeout 300 python bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline --persistent 2 2>&1 | tail -1 | summ
echo "== stats"; timeout 600 python tools/stats_run.py 2>&1 | grep -v amdgpu.ids
