#!/bin/bash
# r02 run 38: kernel timeline of rank 0 of 8 (loopback), 8 in flight
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof38
cd /tmp
TPT_EMU_N=8 TPT_EMU_FRAMES=80 timeout 200 rocprofv3 --kernel-trace --memory-copy-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof38 -o n8 --output-format csv -- python $GRAFT_REPO_ROOT/tools/shard_loopback.py 2>&1 | grep "^N="
ls -la $GRAFT_REPO_ROOT/gpurun_out/prof38
