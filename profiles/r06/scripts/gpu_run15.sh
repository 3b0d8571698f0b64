#!/bin/bash
# Round 6, GPU call 15: second sweep at configs[4]: 8 member gathers in flight + longer pair lists (and fewer paths to make their room).
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
for v in shipped r6_mu8cap256 r6_cap320p784 r6_mu8cap320p784 r6_mu8cap384p752 shipped; do
  echo "== $v"
  if [ $v = shipped ]; then unset TPT_LIB_DIR; else export TPT_LIB_DIR=$PWD/tools/_variants/$v; fi
  for rep in 1 2; do timeout 300 python bench.py --workload c5 --steps 24 --warmup 6 --no-extras --no-cpu-baseline --secondary none 2>&1 | grep -v "$F" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d[k] for k in ('value', 'ms_per_step', 'image_fnv')}, d['config']['lds_bytes_per_block'], d['config']['blocks_per_cu'])"; done
done
