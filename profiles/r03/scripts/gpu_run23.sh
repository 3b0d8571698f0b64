#!/bin/bash
# r03 run 23: stream batching (several frames per launch for streaming callers with small frames): tests, loopback, small configs
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
echo "== stream batching test"; timeout 900 python -m pytest tests/test_gpu_api.py -m gpu -x -q -k "stream_batching" --tb=short 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -12
echo "== full GPU suite"; timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d fpl %.2f parity %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks'], d['config']['frames_per_launch'], d.get('parity_ok')))"; }
for args in "--steps 20 --warmup 5" "--steps 200 --warmup 20" "--workload c1 --steps 20 --warmup 5" "--workload c1 --steps 200 --warmup 20"; do echo "-- $args"; timeout 300 python bench.py --no-cpu-baseline --no-extras $args 2>&1 | tail -1 | summ; echo "   (stream batching off)"; TPT_STREAM_BATCH=0 timeout 300 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 $args 2>&1 | tail -1 | summ; done
echo "== loopback, frame by frame"; TPT_EMU_BATCH=1 TPT_EMU_N=1,2,4,8 TPT_EMU_FRAMES=320 timeout 300 python tools/shard_loopback.py 2>&1 | grep "^N="
echo "== loopback, frame by frame, stream batching off"; TPT_STREAM_BATCH=0 TPT_EMU_BATCH=1 TPT_EMU_N=1,2,4,8 TPT_EMU_FRAMES=320 timeout 300 python tools/shard_loopback.py 2>&1 | grep "^N="
