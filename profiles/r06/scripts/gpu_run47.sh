#!/bin/bash
# Round 6, GPU call 47: super-groups of 16 groups (-DTPT_SUPER=16: half the wave-wide level, a dearer stage B; entry areas 192 / 320 /
# 128) against 8 (the tree, TPT_SUPER made a build parameter, the device unit test rewritten per super-group): A/B at configs[4], the
# grouped parity tests on both builds.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
c5() { for rep in 1 2; do timeout 300 python bench.py --workload c5 --steps 24 --warmup 6 --no-extras --no-cpu-baseline --secondary none 2>&1 | grep -v "$F" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d.get(k) for k in ('value', 'ms_per_step', 'image_fnv')}, d['config']['lds_bytes_per_block'], d['config']['blocks_per_cu'])"; done; }
echo "== C5, super-groups of 8 (working tree)"; c5
echo "== C5, super-groups of 16"; TPT_LIB_DIR=$PWD/tools/_variants/r6_super16 c5
echo "== C5, super-groups of 8 again"; c5
echo "== grouped parity, 8"; timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "stress or config5 or group or cloud or 64_entry" 2>&1 | grep "passed\|failed\|Error" | tail -3
echo "== grouped parity, 16"; TPT_LIB_DIR=$PWD/tools/_variants/r6_super16 timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "stress or config5 or group or cloud or 64_entry" 2>&1 | grep "passed\|failed\|Error" | tail -3
