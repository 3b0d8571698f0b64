#!/bin/bash
# Round 6, GPU call 30: what binds the grouped kernel now -- instruction counts and wait / busy cycles per launch (separate pmc passes), stage profile.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== stages of the dealing, C5 (stats2 build)"; N=6 timeout 300 python tools/stats2_c5.py 2>&1 | grep -v "$F" | tail -13
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
pmc a "--workload c5 --steps 3 --warmup 1" SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pmc b "--workload c5 --steps 3 --warmup 1" SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_ANY
pmc c "--workload c5 --steps 3 --warmup 1" SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY
pmc d "--workload c5 --steps 3 --warmup 1" SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU
pmc e "--workload c5 --steps 3 --warmup 1" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_WAVES
