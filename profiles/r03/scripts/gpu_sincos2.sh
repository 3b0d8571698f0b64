#!/bin/bash
# r03, last 24 GPU-seconds: sincos pair evaluated once per polynomial -- steady state (image hash checked against the oracle off-box), then the driver's command with its own parity leg
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
python3 bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 20 --parity-frames 1 2>/dev/null | tail -1 > gpurun_out/bench_c2_steps200_sincos2.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/bench_c2_driver_cmd_sincos2.json
