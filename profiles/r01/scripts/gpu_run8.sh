#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms occ %d grid %d lds %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['blocks_per_cu'], d['config']['grid_blocks'], d['config']['lds_bytes_per_block']))"; }
for lib in default noslp noslp_p1s p1s; do
  if [ $lib = default ]; then unset TPT_LIB; else export TPT_LIB=tools/_variants/$lib/libtoypathtracer_hip.so; fi
  for args in "--fold 0" "--fold 1" "--persistent 2 --fold 0" "--workload c3 --steps 10 --warmup 2" "--workload c5 --steps 5 --warmup 1"; do
    echo "-- $lib $args"; timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $args 2>&1 | tail -1 | summ
  done
done
