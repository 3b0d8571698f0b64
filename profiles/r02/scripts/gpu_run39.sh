#!/bin/bash
# r02 run 39: is rank 0 of 8 host-bound?  enqueue time per frame + HIP API stats
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 200 python tools/shard_loopback.py 2>&1 | grep "^N="
mkdir -p gpurun_out/prof39
cd /tmp
TPT_EMU_N=8 TPT_EMU_FRAMES=200 timeout 200 rocprofv3 --hip-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof39 -o n8 --output-format csv -- python $GRAFT_REPO_ROOT/tools/shard_loopback.py 2>&1 | grep "^N="
head -30 $GRAFT_REPO_ROOT/gpurun_out/prof39/n8_hip_api_stats.csv
rm -f $GRAFT_REPO_ROOT/gpurun_out/prof39/n8_hip_api_trace.csv
