#!/bin/bash
# Round 6, GPU call 32: positions of the three stages from DPP prefix sums over the lanes instead of returning LDS atomics on one counter, against the committed tree; parity; LDS counters.
# top of the three-stage dealing, against the committed tree (two dealt stages); parity; LDS conflict counters.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
c5() { for rep in 1 2; do timeout 300 python bench.py --workload c5 --steps 24 --warmup 6 --no-extras --no-cpu-baseline --secondary none 2>&1 | grep -v "$F" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d.get(k) for k in ('value', 'ms_per_step', 'image_fnv')}, d['config']['lds_bytes_per_block'], d['config']['blocks_per_cu'])"; done; }
echo "== C5, working tree (prefix sums)"; c5
echo "== C5, committed tree"; TPT_LIB_DIR=$PWD/tools/_variants/r6_big c5
echo "== C5, working tree again"; c5
echo "== grouped parity"; timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "stress or config5 or group" 2>&1 | grep -v "$F" | tail -6
pmc() { # name, bench args, counters...
  local name=$1 args=$2; shift; shift
  cd /tmp && TPT_GRID_DIV=8 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$R/gpurun_out/pmc_r06_$name" -o p -- python "$R/bench.py" $args --no-cpu-baseline --overlap 1 --no-extras --secondary none --parity-frames 0 > /dev/null 2>&1
  cd "$R"; python - <<PY
import csv, glob, collections
for f in glob.glob('gpurun_out/pmc_r06_$name/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'Trace' in r['Kernel_Name']: acc[(r['Kernel_Name'][:44], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()): print('%-46s %-22s mean %18.1f  n %d' % (k[0], k[1], sum(v)/len(v), len(v)))
PY
  rm -rf "$R/gpurun_out/pmc_r06_$name"
}
echo "== PMC C5"
pmc e "--workload c5 --steps 3 --warmup 1" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_ANY
echo "== stages (stats2 build)"; N=6 timeout 300 python tools/stats2_c5.py 2>&1 | grep -v "$F" | tail -13
