#!/bin/bash
# Round 6, GPU call 33: member masks through v_alignbit (one instruction per member instead of compare + select + shift-or) against the
# tree of the second evidence run; grouped parity; pmc of the final grouped kernel (VALU / SALU, FETCH_SIZE, WRITE_SIZE: separate passes).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
c5() { for rep in 1 2; do timeout 300 python bench.py --workload c5 --steps 24 --warmup 6 --no-extras --no-cpu-baseline --secondary none 2>&1 | grep -v "$F" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d.get(k) for k in ('value', 'ms_per_step', 'image_fnv')}, d['config']['lds_bytes_per_block'], d['config']['blocks_per_cu'])"; done; }
echo "== C5, working tree"; c5
echo "== C5, tree of the evidence run"; TPT_LIB_DIR=$PWD/tools/_variants/r6_big c5
echo "== C5, working tree again"; c5
echo "== grouped parity"; timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "stress or config5 or group" 2>&1 | grep -v "$F" | tail -4
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
echo "== PMC C5 (final kernel)"
pmc c5f "--workload c5 --steps 3 --warmup 1" FETCH_SIZE
pmc c5w "--workload c5 --steps 3 --warmup 1" WRITE_SIZE
pmc c5v "--workload c5 --steps 3 --warmup 1" SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pmc c5m "--workload c5 --steps 3 --warmup 1" SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU
pmc c5l "--workload c5 --steps 3 --warmup 1" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU
