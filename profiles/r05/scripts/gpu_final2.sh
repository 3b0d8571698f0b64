#!/bin/bash
# Round 5, after the closing run: the guard added on top (a process that started HIP with GPU_MAX_HW_QUEUES > 22 keeps the groups' bounds off
# the matrix cores; tptGetSceneInfo) -- the tests that touch it, and its effect at C5.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step' % (d['value'], d['ms_per_step']))"; }
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_abi.py -m gpu -q -k "config5 or many_streams or abi or exports" 2>&1 | grep -v "$F" | grep "passed\|failed\|^E   \|Error" | cut -c1-300 | head
echo "== c5, default (20 queues)"; timeout 200 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --workload c5 --steps 40 --warmup 20 2>/dev/null | tail -1 | summ
echo "== c5, host exported GPU_MAX_HW_QUEUES=32 (bounds on the VALU)"; GPU_MAX_HW_QUEUES=32 timeout 200 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --workload c5 --steps 40 --warmup 20 2>/dev/null | tail -1 | summ
echo "== the 4096-sphere frame beside 16 extra streams at 32 queues, guard active: 40 x 2 x 3 frames"
GPU_MAX_HW_QUEUES=32 C5_PATH=device C5_DISTURB=torch_streams timeout 300 python tools/c5_after_hooks.py 40 keep 2>&1 | grep -v "$F" | grep "results\|rror" | tail -2 | cut -c1-600
