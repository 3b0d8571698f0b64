#!/bin/bash
# Round 6, GPU call 46: stream batching (several frames per launch behind a frame-by-frame caller's back, today for frames under 2.4 M
# samples) extended to configs[1]-sized frames: 2 / 4 frames per launch against 1 -- the driver's 20-frame burst, 30 / 100 / 200 frames.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
b() { timeout 300 python3 bench.py --gpus 1 --no-cpu-baseline --no-extras --secondary none "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('%8.1f Mray/s  %.4f ms/frame  steps %d  parity %s golden %s' % (d['value'], d['ms_per_step'], d['steps'], d.get('parity_ok'), (d.get('reference_golden') or {}).get('ok')))"; }
for v in shipped r6_sb2 r6_sb4; do
  if [ $v = shipped ]; then unset TPT_LIB_DIR; else export TPT_LIB_DIR=$PWD/tools/_variants/$v; fi
  echo "== $v"
  b --steps 20 --warmup 5; b --steps 20 --warmup 5
  b --steps 30 --warmup 5 --parity-frames 0; b --steps 100 --warmup 5 --parity-frames 0; b --parity-frames 0
done
