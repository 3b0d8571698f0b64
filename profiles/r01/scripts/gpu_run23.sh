#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
echo "== pytest gpu"; timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== image"; timeout 120 python examples/render_image.py 640 360 64 gpurun_out 2>&1 | grep -v amdgpu
echo "== bench default"; timeout 300 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_c2_default.json | cut -c1-400
echo "== rocprof kernel trace, default command"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_default" -o c2 -- python "$R/bench.py" --no-cpu-baseline > "$R/gpurun_out/prof_default_bench.json" 2>/dev/null
cd "$R"; head -5 gpurun_out/prof_default/c2_kernel_stats.csv; tail -1 gpurun_out/prof_default_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench under rocprof: value', d['value'], 'launch_ms_avg', d['trace_launch_ms_avg'], 'pipeline', d['pipeline_ms_per_step'])"
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c"
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$R/gpurun_out/pmc_$c" -o p -- python "$R/bench.py" --steps 20 --warmup 10 --no-cpu-baseline --overlap 1 > /dev/null 2>&1
  cd "$R"; python - <<PY
import csv, glob, collections
for f in glob.glob('gpurun_out/pmc_$c/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k, v in acc.items(): print(k, 'mean', sum(v)/len(v), 'n', len(v))
PY
done
bash tools/gpu_pmc.sh "--overlap 1 --steps 20 --warmup 10" final 2>&1 | grep -v amdgpu.ids
