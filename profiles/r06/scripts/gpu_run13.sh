#!/bin/bash
# Round 6, GPU call 13: one multi-GPU implementation (bench at N = 1 on the plain device path; the C-ABI exchange with its new exchange
# interval for small tiles): loopback table with the automatic interval and with an exchange every frame, the GPU tests of the sharded
# path, the driver's command.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== loopback, automatic exchange interval"; timeout 600 python tools/shard_loopback.py 2>&1 | grep -v "$F" | tail -5
echo "== loopback, exchange every frame"; TPT_EMU_EVERY=1 timeout 600 python tools/shard_loopback.py 2>&1 | grep -v "$F" | tail -5
echo "== loopback at C3 (3840x2160x16: exchange every frame by the automatic rule)"; TPT_EMU_SIZE=3840x2160 TPT_EMU_FRAMES=40 timeout 600 python - <<'PY' 2>&1 | grep -v "$F" | tail -5
import os, runpy, sys
sys.argv = ["tools/shard_loopback.py"]
from toypathtracer_amd import api
api.InitializeTest(); api.set_samples_per_pixel(16)
runpy.run_path("tools/shard_loopback.py", run_name="__main__")
PY
timeout 1500 python -m pytest tests/test_gpu_api.py -x -q 2>&1 | grep -v "$F" | tail -6
echo "== the driver's command"
for rep in 1 2; do timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --secondary none 2>&1 | grep -v "$F" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d.get(k) for k in ('value', 'ms_per_step', 'parity_ok', 'image_fnv', 'exchange')}, d['reference_golden']['ok'])"; done
echo "== 200 frames"
timeout 600 python bench.py --no-cpu-baseline --secondary none --no-extras 2>&1 | grep -v "$F" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d.get(k) for k in ('value', 'ms_per_step', 'image_fnv')}, d['reference_golden']['ok'])"
