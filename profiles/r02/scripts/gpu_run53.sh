#!/bin/bash
# r02 run 53: torch sharding path (tools/shard_exchange_emu.py): why was N=4 slow (0.379 ms/frame) in run 52?
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
run() { echo "== $*"; env "$@" TPT_EMU_FRAMES=300 timeout 200 python tools/shard_exchange_emu.py 2>&1 | grep "^N="; }
run TPT_EMU_N=4
run TPT_EMU_N=4
run TPT_EMU_N=2,4
run TPT_EMU_N=4 TPT_HOST_PACE=0
run TPT_EMU_N=4 TPT_SLOT_FACTOR=1
run TPT_EMU_N=4 TPT_EMU_MIRROR=0
run TPT_EMU_N=3,5,6
