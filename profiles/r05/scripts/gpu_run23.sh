#!/bin/bash
# Round 5, GPU call 23: the mitigation candidate -- the process never holds more hardware queues than the device runs side by side
# (call 19: 20 clean, 24 not).  What does GPU_MAX_HW_QUEUES=20 / 22 cost the headline, and is 22 still clean with a second context?
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  hw_queues %s' % (d['value'], d['ms_per_step'], d.get('pipeline', {}).get('hw_queues')))"; }
for q in 32 20 22; do
  echo "== GPU_MAX_HW_QUEUES=$q: driver's command x2, 200 steps x1"
  for i in 1 2; do GPU_MAX_HW_QUEUES=$q timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --parity-frames 0 2>/dev/null | tail -1 | summ; done
  GPU_MAX_HW_QUEUES=$q timeout 200 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-extras --parity-frames 0 2>/dev/null | tail -1 | summ
done
echo "== second context initialised, GPU_MAX_HW_QUEUES=22"
GPU_MAX_HW_QUEUES=22 C5_PATH=device C5_DISTURB=hooks_init timeout 300 python tools/c5_after_hooks.py 40 keep 2>&1 | grep -v "$F" | grep "results\|rror" | tail -3
