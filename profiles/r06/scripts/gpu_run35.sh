#!/bin/bash
# Round 6, GPU call 35: the overflow paths of the three-stage dealing.  A build whose entry areas hold 64 entries each (TPT_DEAL_CA = CB =
# CS = 64; the flat variants' pair list 128) -- super-group entries spill into further rounds, group entries and survivors that find their
# stack full are served in place by the lane holding them -- must render the same bits: grouped parity tests + image hash of 30 C5 frames.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
export TPT_LIB_DIR=$PWD/tools/_variants/r6_tiny
echo "== C5, 64-entry areas"; timeout 300 python bench.py --workload c5 --steps 24 --warmup 6 --no-extras --no-cpu-baseline --secondary none 2>&1 | grep -v "$F" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d.get(k) for k in ('value', 'ms_per_step', 'image_fnv')}, d['config']['lds_bytes_per_block'], d['config']['blocks_per_cu'])"
echo "== grouped parity, 64-entry areas"; timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "stress or config5 or group or both_kernels" 2>&1 | grep -v "$F" | tail -5
