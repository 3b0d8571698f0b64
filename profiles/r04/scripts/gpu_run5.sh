#!/bin/bash
# r04 run 5: which test of the quick parity subset kills the process (run 4: "dumped core" for every build)?
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
timeout 900 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -v -x -k "per_pixel_bit_exact or config2_1280 or overlap_is or animated or small_scenes or spp or golden" 2>&1 | grep -v "$F" | tail -60
dmesg 2>/dev/null | tail -5
