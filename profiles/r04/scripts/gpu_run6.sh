#!/bin/bash
# r04 run 6: isolate the abort in test_per_pixel_bit_exact_all_variants (alone / after the row-serial tests), with the runtime's error log
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== alone"
AMD_LOG_LEVEL=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "per_pixel_bit_exact" 2>&1 | grep -v "$F" | grep -v "^\s*File\|^Extension" | tail -12
echo "== after the batched row-serial tests"
AMD_LOG_LEVEL=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "row_serial_batched_launch or per_pixel_bit_exact" 2>&1 | grep -v "$F" | grep -v "^\s*File\|^Extension" | tail -12
echo "== after the golden row-serial tests"
AMD_LOG_LEVEL=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "row_serial_reproduces or per_pixel_bit_exact" 2>&1 | grep -v "$F" | grep -v "^\s*File\|^Extension" | tail -12
