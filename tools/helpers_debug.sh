#!/bin/bash
# Next round, first GPU call: run down the order-dependent failure of test_sharded_batches_match_per_frame_exchange on the
# -DTPT_TAIL_HELPERS=1 build (profiles/r04/r04_run24.log: 20 tests of tests/test_gpu_api.py green, then this one red; green alone).
#   TPT_EXTRA_FLAGS="-DTPT_TAIL_HELPERS=1" TPT_OUT_DIR=$PWD/tools/_variants/helpers bash toypathtracer_amd/csrc/build.sh   (here, before gpurun)
#   gpurun --timeout 600 -- 'bash tools/helpers_debug.sh > gpurun_out/helpers_debug.log 2>&1; tail -60 gpurun_out/helpers_debug.log'
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
V=$PWD/tools/_variants/helpers
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
T=tests/test_gpu_api.py
bad=test_sharded_batches_match_per_frame_exchange
echo "== the file in order, helpers on";            TPT_LIB_DIR=$V timeout 200 python -m pytest $T -m gpu -x -q 2>&1 | grep -v "$F" | tail -4
echo "== the file in order, helpers switched off";  TPT_LIB_DIR=$V TPT_TAIL_HELPERS=0 timeout 200 python -m pytest $T -m gpu -x -q 2>&1 | grep -v "$F" | tail -4
echo "== the file in order, shipped library, 3x";   for i in 1 2 3; do timeout 200 python -m pytest $T -m gpu -x -q 2>&1 | grep -v "$F" | tail -1; done
# which predecessor matters: each earlier test of the file + the failing one
for t in $(grep -o '^def test_[a-z_0-9]*' $T | sed 's/def //' | grep -v $bad); do
  r=$(TPT_LIB_DIR=$V timeout 120 python -m pytest $T -m gpu -q -x -k "$t or $bad" 2>&1 | grep -v "$F" | tail -1)
  echo "-- $t + $bad: $r"
done
