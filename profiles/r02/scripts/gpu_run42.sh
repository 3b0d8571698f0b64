#!/bin/bash
# r02 run 42: two colour slots per trace stream (blend chain off the streams' critical path), A/B
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks']))"; }
for sf in 1 2; do for args in "--steps 20 --warmup 5" "--steps 200 --warmup 20" "--steps 600 --warmup 20" "--workload c1 --steps 400 --warmup 40" "--workload c3 --steps 60 --warmup 10" "--animate"; do echo "-- TPT_SLOT_FACTOR=$sf $args"; TPT_SLOT_FACTOR=$sf timeout 300 python bench.py --no-cpu-baseline --no-extras $args 2>/dev/null | tail -1 | summ; done; done
for sf in 1 2; do echo "== TPT_SLOT_FACTOR=$sf loopback"; TPT_SLOT_FACTOR=$sf timeout 200 python tools/shard_loopback.py 2>&1 | grep "^N="; done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -q -x -m gpu -k "overlap or golden or pipelined or animated or loopback or lookahead or stress" 2>&1 | tail -3
