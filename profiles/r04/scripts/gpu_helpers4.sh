#!/bin/bash
# r04: tail helpers, third form (helpers go to the streams that have nothing queued behind their launch); second form was: no streams of their own (the k-th newest launch is helped from the stream of the k-th oldest), the
# hand-shake through dependent returning atomics instead of seq_cst fences.  A/B against the shipped library, the per-frame overhead
# alone (TPT_HELPER_MAX=0: serial, event, closing hand-shake, no helper launches), a timeline, then as much of the GPU suite as fits.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
mkdir -p gpurun_out/tr
Q="--no-cpu-baseline --no-extras --parity-frames 0"
V=$R/tools/_variants/helpers
val() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f %.4f launch %.3f %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d.get('parity_ok')))"; }
run() { timeout 60 python bench.py --gpus 1 "$@" 2>/dev/null | grep '^{"metric"' | tail -1 | val; }
timeout 200 python -c "import torch; print(torch.cuda.get_device_name(0))" 2>/dev/null
echo "== helpers on, the driver's command with the oracle leg"; P=$(TPT_LIB_DIR=$V run --steps 20 --warmup 5 --no-cpu-baseline --no-extras); echo "$P"
for steps in 20 30 200; do
  W=5; [ $steps = 200 ] && W=20
  B=$(run --steps $steps --warmup $W $Q); echo "-- $steps: shipped                  $B"
  echo "-- $steps: variant, overhead only    $(TPT_LIB_DIR=$V TPT_HELPER_MAX=0 run --steps $steps --warmup $W $Q)"
  H=$(TPT_LIB_DIR=$V run --steps $steps --warmup $W $Q); echo "-- $steps: variant, helpers on      $H"
  [ $steps = 20 ] && B20=$B && H20=$H
done
echo "-- 20 again: shipped / on / on, 4 launches   $(run --steps 20 --warmup 5 $Q) / $(TPT_LIB_DIR=$V run --steps 20 --warmup 5 $Q) / $(TPT_LIB_DIR=$V TPT_HELPER_MAX=4 run --steps 20 --warmup 5 $Q)"
echo "elapsed $SECONDS s"
echo "== trace: 20 steps, helpers on"; (cd /tmp && TPT_LIB_DIR=$V timeout 60 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr/s20h3 -o t -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 $Q 2>/dev/null | grep '^{"metric"' | tail -1 | val)
find gpurun_out/tr/s20h3 -name "*.csv" ! -name "*kernel_trace.csv" -delete
GAIN=$(python -c "b=float('$B20'.split()[0]); h=float('$H20'.split()[0]); print(1 if h > 1.03 * b else 0)")
OK=$(python -c "print(1 if '$P'.split()[-1] == 'True' else 0)")
echo "gain $GAIN parity $OK elapsed $SECONDS s"
if [ "$GAIN" = 1 ] && [ "$OK" = 1 ]; then
  LEFT=$((112 - SECONDS)); echo "== GPU tests on the variant ($LEFT s)"
  TPT_LIB_DIR=$V timeout $LEFT python -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py -m gpu -x -q --timeout=100 2>&1 | grep -v 'RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids' | tail -6
fi
echo "elapsed $SECONDS s"
