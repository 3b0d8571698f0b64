#!/bin/bash
# usage: tools/gpu_pmc.sh "<bench args>" TAG
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="$1"; TAG=$2
run() { # name counters...
  local name=$1; shift
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$R/gpurun_out/pmc_${TAG}_$name" -o p -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline $ARGS > /dev/null 2>&1
  cd "$R"; python - <<PY
import csv, glob, collections
for f in glob.glob('gpurun_out/pmc_${TAG}_$name/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'Trace' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()): print('%-28s mean %16.1f  n %d' % (k, sum(v)/len(v), len(v)))
PY
}
echo "== PMC $TAG: $ARGS"
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE
run b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT
run c SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS SQ_INSTS_BRANCH
