#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
echo "== stats"; timeout 600 python tools/stats_run.py 2>&1 | grep -v amdgpu.ids
echo "== rocprof kernel trace c2 (csv)"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_c2" -o c2 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"; find gpurun_out/prof_c2 -name "*.csv" | head; cat $(find gpurun_out/prof_c2 -name "*kernel_stats.csv" | head -1) | head -5
echo "== pmc pass 1 (SQ)"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc1" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"; python - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/pmc1/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'tptTraceKernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items(): print(k, sum(v)/len(v), len(v))
PY
echo "== pmc pass 2"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc2" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"; python - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/pmc2/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'tptTraceKernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items(): print(k, sum(v)/len(v), len(v))
PY
