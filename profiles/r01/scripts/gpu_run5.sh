#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
echo "== bench default"; timeout 600 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_c2_default.json
echo "== bench overlap 1"; timeout 600 python bench.py --overlap 1 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_c2_overlap1.json
echo "== rocprof kernel trace, default command"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_default" -o c2 -- python "$R/bench.py" --no-cpu-baseline > "$R/gpurun_out/prof_default_bench.json" 2>/dev/null
cd "$R"; cat gpurun_out/prof_default/c2_kernel_stats.csv | head -4; tail -1 gpurun_out/prof_default_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench under rocprof: value', d['value'], 'launch_ms_avg', d['trace_launch_ms_avg'], 'pipeline', d['pipeline_ms_per_step'])"
echo "== rocprof kernel trace, --overlap 1"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_overlap1" -o c2 -- python "$R/bench.py" --no-cpu-baseline --overlap 1 > "$R/gpurun_out/prof_overlap1_bench.json" 2>/dev/null
cd "$R"; cat gpurun_out/prof_overlap1/c2_kernel_stats.csv | head -4; tail -1 gpurun_out/prof_overlap1_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench under rocprof: value', d['value'], 'launch_ms_avg', d['trace_launch_ms_avg'], 'pipeline', d['pipeline_ms_per_step'])"
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c"
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$R/gpurun_out/pmc_$c" -o p -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --overlap 1 > /dev/null 2>&1
  cd "$R"; python - <<PY
import csv, glob, collections
for f in glob.glob('gpurun_out/pmc_$c/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k, v in acc.items(): print(k, 'mean', sum(v)/len(v), 'n', len(v))
PY
done
echo "== pmc SQ (overlap 1)"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d "$R/gpurun_out/pmc_sq" -o p -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --overlap 1 > /dev/null 2>&1
cd "$R"; python - <<PY
import csv, glob, collections
for f in glob.glob('gpurun_out/pmc_sq/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'Trace' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items(): print(k, 'mean', sum(v)/len(v), 'n', len(v))
PY
