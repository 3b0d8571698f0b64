#!/bin/bash
# r04 run 16: members per group 4 / 6 / 8 / 12 / 16 on the 4096-sphere scene (the group bounds are cheap on the matrix cores now)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks']))"; }
for v in g4 g6 g8 g12 base g8 g6; do
  if [ $v = base ]; then unset TPT_LIB; else export TPT_LIB=$R/tools/_variants/$v/libtoypathtracer_hip.so; fi
  echo "== [$v] c5"; timeout 120 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --workload c5 --steps 40 --warmup 20 2>&1 | tail -1 | summ
done
