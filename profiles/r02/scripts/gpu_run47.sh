#!/bin/bash
# r02 run 47: several frames per launch (tptDrawDeviceBatch / tptDrawShardedBatch): parity, rates
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -q -x -m gpu -k "batch" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -15
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d  batched4 %s  batched8 %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks'], d.get('batched_4_Mray_s'), d.get('batched_8_Mray_s')))"; }
for args in "--steps 200 --warmup 20" "--workload c1 --steps 400 --warmup 40" "--workload c3 --steps 40 --warmup 10"; do echo "-- $args"; timeout 300 python bench.py --no-cpu-baseline $args 2>/dev/null | tail -1 | summ; done
for b in 1 2 4 8 16; do echo "== loopback, batch $b"; TPT_EMU_BATCH=$b TPT_EMU_FRAMES=320 timeout 200 python tools/shard_loopback.py 2>&1 | grep "^N="; done
