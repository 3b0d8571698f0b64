#!/bin/bash
# Round 6, GPU call 17: WHICH instruction class poisons the later gathers?  The traversal runs on VALU masks; beside it, as dead work:
#   d4u = operands + lane swaps + A loads + 32 MFMAs + sign extraction (everything, unrolled like the product was)
#   d2noswap = A loads + 32 MFMAs + sign extraction, NO v_permlane32_swap anywhere
#   d3heavy = 48 v_permlane32_swap per call, nothing else        d1 = the A loads only
# 300 sets of 3 frames each, 32 queues + 16 streams.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
for v in r6_d4u r6_d2noswap r6_d3heavy r6_d1; do
  echo "== $v"
  C5_LIB_SEES=20 TPT_LIB_DIR=$PWD/tools/_variants/$v timeout 900 python tools/c5_timeslice.py 300 3 2>&1 | grep -v "$F" | grep "c5_timeslice:" | cut -c1-200
done
echo "== loopback with deferred batching of small sharded tiles (automatic) / every frame"
timeout 600 python tools/shard_loopback.py 2>&1 | grep -v "$F" | tail -4
TPT_EMU_EVERY=1 timeout 600 python tools/shard_loopback.py 2>&1 | grep -v "$F" | tail -4
timeout 900 python -m pytest tests/test_gpu_api.py -x -q 2>&1 | grep -v "$F" | tail -4
