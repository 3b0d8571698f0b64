#!/bin/bash
# r02 run 54: how often does a timed pass of a sharded-tile run contain a one-off stall?  C ABI (loopback) vs torch path
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
echo "== C ABI loopback"; TPT_EMU_VERBOSE=1 TPT_EMU_N=3,4,5,6,4,6,3,5 TPT_EMU_FRAMES=300 timeout 200 python tools/shard_loopback.py 2>&1 | grep "pass"
echo "== torch path"; TPT_EMU_VERBOSE=1 TPT_EMU_N=3,4,5,6,4,6,3,5 TPT_EMU_FRAMES=300 timeout 200 python tools/shard_exchange_emu.py 2>&1 | grep "pass"
