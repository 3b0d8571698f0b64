#!/bin/bash
# Round 6, GPU call 36: the half-line test on the bounds (super-groups wave-wide, groups in stage B: a bound whose centre is behind the ray's
# origin and which the origin is outside of is dropped) against the line test only (-DTPT_DEAL_HALF_LINE=0, same tree); the device unit
# test of the bounds filters (now also: half-line masks against the reference's whole acceptance), grouped parity incl. the new 64-entry
# overflow test and the time-sliced child; stage profile.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
c5() { for rep in 1 2; do timeout 300 python bench.py --workload c5 --steps 24 --warmup 6 --no-extras --no-cpu-baseline --secondary none 2>&1 | grep -v "$F" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d.get(k) for k in ('value', 'ms_per_step', 'image_fnv')}, d['config']['lds_bytes_per_block'], d['config']['blocks_per_cu'])"; done; }
echo "== C5, working tree (half-line test)"; c5
echo "== C5, line test only"; TPT_LIB_DIR=$PWD/tools/_variants/r6_nohalf c5
echo "== C5, working tree again"; c5
echo "== grouped parity"; timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -k "stress or config5 or group or time_sliced or both_kernels" 2>&1 | grep -v "$F" | tail -6
echo "== stages (stats2 build)"; N=6 timeout 300 python tools/stats2_c5.py 2>&1 | grep -v "$F" | tail -14
