#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms occ %d grid %d lds %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['blocks_per_cu'], d['config']['grid_blocks'], d['config']['lds_bytes_per_block']))"; }
for lib in default noprio; do
  if [ $lib = default ]; then unset TPT_LIB; else export TPT_LIB=tools/_variants/$lib/libtoypathtracer_hip.so; fi
  for args in "" "--overlap 1" "--fold 1" "--workload c3 --steps 20 --warmup 8"; do echo "-- $lib $args"; timeout 90 python bench.py --no-cpu-baseline $args 2>&1 | tail -1 | summ; done
  echo "-- $lib shard emulation"; timeout 100 python profiles/r01/scripts/shard_emulation.py 2>&1 | grep -v amdgpu | grep "rank 0"
done
