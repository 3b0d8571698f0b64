#!/bin/bash
# r02 run 37: loopback communicator (rank 0 of N on one GPU through the C ABI), in-flight depth
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 300 python -m pytest tests/test_gpu_api.py -q -x -m gpu -k "loopback or sharded or shard" 2>&1 | tail -3
echo "== default"; timeout 200 python tools/shard_loopback.py 2>&1 | grep "^N="
for ov in 4 8 12; do echo "== TPT_EMU_OV=$ov"; TPT_EMU_OV=$ov TPT_EMU_N=4,8 timeout 200 python tools/shard_loopback.py 2>&1 | grep "^N="; done
echo "== python sharding path (torch plumbing), 8 in flight"; TPT_EMU_OV=8 TPT_EMU_N=4,8 TPT_EMU_FRAMES=300 timeout 200 python tools/shard_exchange_emu.py 2>&1 | grep "^N="
