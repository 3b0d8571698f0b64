#!/bin/bash
# r04: the tail-helper experiment (-DTPT_TAIL_HELPERS=1, tools/_variants/helpers = csrc/build.sh with that flag: product + hooks builds).
# Parity of the driver's command with the helpers on, A/B against the shipped library on bursts of 20 / 30 / 200 frames, a kernel
# timeline, and -- if the burst gains -- the full GPU suite on the variant.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
mkdir -p gpurun_out/tr
Q="--no-cpu-baseline --no-extras --parity-frames 0"
V=$R/tools/_variants/helpers
val() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f %.4f %s %s' % (d['value'], d['ms_per_step'], d['config'].get('grid_blocks'), d.get('parity_ok')))"; }
run() { timeout 60 python bench.py --gpus 1 "$@" 2>/dev/null | grep '^{"metric"' | tail -1 | val; }
timeout 200 python -c "import torch; print(torch.cuda.get_device_name(0))" 2>/dev/null
echo "== helpers on, the driver's command with the oracle leg"; P=$(TPT_LIB_DIR=$V run --steps 20 --warmup 5 --no-cpu-baseline --no-extras); echo "$P"
for steps in 20 30 200; do
  W=5; [ $steps = 200 ] && W=20
  B=$(run --steps $steps --warmup $W $Q); echo "-- $steps steps: shipped library        $B"
  O=$(TPT_LIB_DIR=$V TPT_TAIL_HELPERS=0 run --steps $steps --warmup $W $Q); echo "-- $steps steps: variant, helpers off    $O"
  H=$(TPT_LIB_DIR=$V run --steps $steps --warmup $W $Q); echo "-- $steps steps: variant, helpers on     $H"
  [ $steps = 20 ] && B20=$B && H20=$H
done
echo "-- 20 steps: helpers on, up to 4 launches  $(TPT_LIB_DIR=$V TPT_HELPER_MAX=4 run --steps 20 --warmup 5 $Q)"
echo "-- 20 steps: helpers on, up to 16, pct 10  $(TPT_LIB_DIR=$V TPT_HELPER_MAX=16 TPT_HELPER_PCT=10 run --steps 20 --warmup 5 $Q)"
echo "-- 20 steps again: shipped / helpers on    $(run --steps 20 --warmup 5 $Q) / $(TPT_LIB_DIR=$V run --steps 20 --warmup 5 $Q)"
echo "elapsed $SECONDS s"
echo "== trace: 20 steps, helpers on"; (cd /tmp && TPT_LIB_DIR=$V timeout 90 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr/s20h -o t -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 $Q 2>/dev/null | grep '^{"metric"' | tail -1 | val)
find gpurun_out/tr/s20h -name "*.csv" ! -name "*kernel_trace.csv" -delete
GAIN=$(python -c "b=float('$B20'.split()[0]); h=float('$H20'.split()[0]); print(1 if h > 1.03 * b else 0)")
OK=$(python -c "print(1 if '$P'.split()[-1] == 'True' else 0)")
echo "gain $GAIN parity $OK elapsed $SECONDS s"
if [ "$GAIN" = 1 ] && [ "$OK" = 1 ] && [ $SECONDS -lt 75 ]; then
  echo "== full GPU suite on the variant"; TPT_LIB_DIR=$V timeout 172 python -m pytest tests -m gpu -x -q --timeout=120 2>&1 | grep -v 'RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids' | tail -6
fi
echo "elapsed $SECONDS s"
