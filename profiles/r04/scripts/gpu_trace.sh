#!/bin/bash
# r04 last GPU call (data for the round-5 plan, nothing here changes the product):
#  (1) the late-join experiment (tools/_variants/latejoin = -DTPT_LATE_JOIN=1, tpt_device.h): surplus workgroups that leave when they
#      arrive late, against the plain 200 % / 300 % fill rules, on bursts of 20 / 30 / 200 frames, one of them with the oracle leg;
#  (2) kernel timelines of short bursts (how 16 overlapped launches tile the 512 workgroup slots: ramp, rounds, half-empty tail);
#  (3) the driver's command once more at HEAD.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
mkdir -p gpurun_out/tr
Q="--no-cpu-baseline --no-extras --parity-frames 0"
V=$R/tools/_variants/latejoin/libtoypathtracer_hip.so
pick() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f Mray/s %.4f ms/step launch %.3f ms grid %s parity %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config'].get('grid_blocks'), d.get('parity_ok')))"; }
run() { timeout 60 python bench.py --gpus 1 "$@" 2>/dev/null | grep '^{"metric"' | tail -1 | pick; }
timeout 200 python -c "import torch; print(torch.cuda.get_device_name(0))" 2>/dev/null; echo "import done at $SECONDS s"
echo "== late join, with the oracle leg (driver's command, pct 75)"; TPT_LIB=$V TPT_JOIN_PCT=75 run --steps 20 --warmup 5 --no-cpu-baseline --no-extras
for steps in 20 30 200; do
  W=5; [ $steps = 200 ] && W=20
  echo "-- $steps steps: base (fill 200 %)"; run --steps $steps --warmup $W $Q
  echo "-- $steps steps: fill 300 %"; TPT_GRID_FILL=300 run --steps $steps --warmup $W $Q
  for pct in 50 75 90; do echo "-- $steps steps: late join x2, pct $pct"; TPT_LIB=$V TPT_JOIN_PCT=$pct run --steps $steps --warmup $W $Q; done
  echo "-- $steps steps: late join x4, pct 75"; TPT_LIB=$V TPT_JOIN_PCT=75 TPT_JOIN_MULT=4 run --steps $steps --warmup $W $Q
done
echo "elapsed $SECONDS s"
tr() { # name, env..., args
  local name=$1; shift
  (cd /tmp && env "$@" timeout 90 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr/$name -o t -- python $R/bench.py --gpus 1 --warmup 5 $Q $TRARGS 2>/dev/null | grep '^{"metric"' | tail -1 | pick)
}
if [ $SECONDS -lt 200 ]; then echo "== trace: 20 steps"; TRARGS="--steps 20" tr s20 A=1; fi
if [ $SECONDS -lt 215 ]; then echo "== trace: 20 steps, late join x2 pct 75"; TRARGS="--steps 20" tr s20lj TPT_LIB=$V TPT_JOIN_PCT=75; fi
if [ $SECONDS -lt 230 ]; then echo "== trace: 30 steps"; TRARGS="--steps 30" tr s30 A=1; fi
find gpurun_out/tr -name "*.csv" ! -name "*kernel_trace.csv" -delete
if [ $SECONDS -lt 250 ]; then echo "== driver's command at HEAD"; timeout 60 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{"metric"' | tail -1 | tee gpurun_out/tr/bench_c2_driver_cmd_head.json | pick; fi
echo "elapsed $SECONDS s"
