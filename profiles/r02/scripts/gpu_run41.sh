#!/bin/bash
# r02 run 41: paced host, sharded tiles: grid fill x frames in flight
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for env in "TPT_GRID_FILL=100 TPT_SHARD_CAP=8" "TPT_GRID_FILL=150 TPT_SHARD_CAP=8" "TPT_GRID_FILL=200 TPT_SHARD_CAP=8" "TPT_GRID_FILL=400 TPT_SHARD_CAP=8" "TPT_GRID_FILL=200 TPT_SHARD_CAP=6" "TPT_GRID_FILL=200 TPT_SHARD_CAP=12" "TPT_GRID_FILL=50 TPT_SHARD_CAP=8"; do
  echo "== $env"; env TPT_HOST_PACE=1 $env TPT_EMU_N=4,8 timeout 200 python tools/shard_loopback.py 2>&1 | grep "^N="
done
