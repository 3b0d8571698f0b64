#!/bin/bash
# Round 6, GPU call 25: the next sub-round's member gathers in flight during this sub-round's push / exact pass: 0 (the committed tree), 4
# or 8 of the 8 members prefetched (16 / 26 spilled VGPRs).  Image hash of 30 frames must stay a29a64af.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
c5() { for rep in 1 2; do timeout 300 python bench.py --workload c5 --steps 24 --warmup 6 --no-extras --no-cpu-baseline --secondary none 2>&1 | grep -v "$F" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d.get(k) for k in ('value', 'ms_per_step', 'image_fnv')}, d['config']['lds_bytes_per_block'], d['config']['blocks_per_cu'])"; done; }
for P in 0 4 8 0; do echo "== prefetch $P"; TPT_LIB_DIR=$PWD/tools/_variants/r6_pf$P c5; done
