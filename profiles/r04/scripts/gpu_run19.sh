#!/bin/bash
# r04 run 19: member records requested together in the member filter (TPT_MEMBER_UNROLL 2 / 4 / 8) at C5, 300 frames each
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.3f ms/step  launch %.1f ms  image %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d.get('image_fnv')))"; }
for v in base mu8 mu2 base mu8; do
  if [ $v = base ]; then unset TPT_LIB; else export TPT_LIB=$R/tools/_variants/$v/libtoypathtracer_hip.so; fi
  echo "== [$v] c5 x 300"; timeout 100 python bench.py --no-cpu-baseline --no-extras --parity-frames 1 --workload c5 --steps 300 --warmup 20 2>/dev/null | grep '^{"metric"' | tail -1 | line
done
