#!/bin/bash
# Round 6, GPU call 18: the working tree after the one-pass two-level dealing (dealTwoLevel: bounds filter + producer side of the dealing
# per chunk of 512 groups) and the deferred batching of small sharded frames: grouped parity, C5 rate against the committed tree (A/B in
# one call), API + sharding tests, loopback table.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
c5() { for rep in 1 2; do timeout 300 python bench.py --workload c5 --steps 24 --warmup 6 --no-extras --no-cpu-baseline --secondary none 2>&1 | grep -v "$F" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d[k] for k in ('value', 'ms_per_step', 'image_fnv', 'parity_ok')}, d['config']['lds_bytes_per_block'], d['config']['blocks_per_cu'])"; done; }
echo "== C5, working tree"; c5
echo "== C5, committed tree (8a9c282)"; TPT_LIB_DIR=$PWD/tools/_variants/r6_head c5
echo "== C5, working tree again"; c5
echo "== parity (whole file)"; timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | grep -v "$F" | tail -5
echo "== api"; timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_zz_lifecycle.py -x -q 2>&1 | grep -v "$F" | tail -4
echo "== loopback"; timeout 600 python tools/shard_loopback.py 2>&1 | grep -v "$F" | tail -4
