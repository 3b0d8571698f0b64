#!/bin/bash
# Round 5, GPU call 4: the whole of tests/test_gpu_api.py in order (subprocess tests included: without them the helpers build passed
# 4 of 4, call 3), helpers build x4 then shipped build x3, with the mismatch diagnostics of the sharded tests.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
V=$PWD/tools/_variants/helpers
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
t0=$(date +%s)
for i in 1 2 3 4; do
  echo "== helpers build, run $i"; TPT_LIB_DIR=$V timeout 300 python -m pytest tests/test_gpu_api.py -m gpu -q 2>&1 | grep -v "$F" | grep -v "^$" | tail -45 | cut -c1-300
done
for i in 1 2 3; do
  echo "== shipped build, run $i"; timeout 300 python -m pytest tests/test_gpu_api.py -m gpu -q 2>&1 | grep -v "$F" | grep -v "^$" | tail -30 | cut -c1-300
done
echo "elapsed $(( $(date +%s) - t0 )) s"
