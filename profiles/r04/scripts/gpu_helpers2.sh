#!/bin/bash
# r04: why is the -DTPT_TAIL_HELPERS=1 build slow even with its helpers switched off (r04_run21.log)?  The build differs from the
# shipped one in the kernel's code (prologue / epilogue branches, 14 more spilled SGPRs), the stack stride (room for the helper
# columns) and -- in run 21 -- four more streams created at start-up (now created on first use).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
Q="--no-cpu-baseline --no-extras --parity-frames 0"
V=$R/tools/_variants/helpers
val() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f %.4f launch %.3f' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg']))"; }
run() { timeout 60 python bench.py --gpus 1 "$@" 2>/dev/null | grep '^{"metric"' | tail -1 | val; }
timeout 200 python -c "import torch; print(torch.cuda.get_device_name(0))" 2>/dev/null
for steps in 200 30; do
  W=5; [ $steps = 200 ] && W=20
  echo "-- $steps: shipped                         $(run --steps $steps --warmup $W $Q)"
  echo "-- $steps: variant, off, plain stride      $(TPT_LIB_DIR=$V TPT_TAIL_HELPERS=0 TPT_HELPER_STRIDE=0 run --steps $steps --warmup $W $Q)"
  echo "-- $steps: variant, off, helper stride     $(TPT_LIB_DIR=$V TPT_TAIL_HELPERS=0 run --steps $steps --warmup $W $Q)"
  echo "-- $steps: variant, on                     $(TPT_LIB_DIR=$V run --steps $steps --warmup $W $Q)"
done
echo "-- 20: shipped / variant on               $(run --steps 20 --warmup 5 $Q) / $(TPT_LIB_DIR=$V run --steps 20 --warmup 5 $Q)"
echo "elapsed $SECONDS s"
