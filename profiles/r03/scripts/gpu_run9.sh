#!/bin/bash
# r03 run 9: is the blend kernel's 40-us-quantised duration tied to trace workgroups that cannot be placed (oversubscribed grids)?
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
for fill in 100 150 200; do
  echo "=== TPT_GRID_FILL=$fill"
  export TPT_GRID_FILL=$fill
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/fill_$fill" -o c2 -- python3 "$R/bench.py" --steps 200 --warmup 20 --no-cpu-baseline --no-extras --parity-frames 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  bench %8.1f Mray/s grid %d' % (d['value'], d['config']['grid_blocks']))"
  cd "$R"; python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/fill_$fill/c2_kernel_trace.csv')))
d=sorted(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows if 'Resolve' in r['Kernel_Name'])
print('  resolve n %d mean %.1f us median %.1f p90 %.1f max %.1f' % (len(d), sum(d)/len(d)/1e3, d[len(d)//2]/1e3, d[int(.9*len(d))]/1e3, d[-1]/1e3))
PY
  timeout 100 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --parity-frames 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  bench (no profiler) %8.1f Mray/s' % d['value'])"
done
