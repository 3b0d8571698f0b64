#!/bin/bash
# usage: tools/traffic.sh "<bench args>"  -- HBM traffic of the trace kernel per launch: FETCH_SIZE / WRITE_SIZE in separate passes
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$R/gpurun_out/pmc_$c" -o p -- python "$R/bench.py" --steps 20 --warmup 10 --no-cpu-baseline --overlap 1 $1 > /dev/null 2>&1
  cd "$R"; python - <<PY
import csv, glob, collections
for f in glob.glob('gpurun_out/pmc_$c/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'tpt' in r['Kernel_Name']: acc[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k, v in acc.items(): print(k, 'mean', sum(v)/len(v), 'n', len(v))
PY
done
