#!/bin/bash
# Round 6, GPU call 12: the product without the matrix-core path in its grouped instantiation (0 spilled VGPRs, no scratch):
# C5 rate, HBM traffic and executed VALU instructions per C5 launch (separate --pmc passes), the driver's command with the new
# secondary legs, the default command, and the GPU tests touched since call 11.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
for rep in 1 2; do
  echo "== bench c5 (rep $rep)"
  timeout 300 python bench.py --workload c5 --steps 24 --warmup 6 --no-extras --no-cpu-baseline --secondary none 2>&1 | grep -v "$F" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d[k] for k in ('value', 'ms_per_step', 'trace_launch_ms_avg', 'image_fnv')}, d['config']['hit_spheres'], d['config']['lds_bytes_per_block'], d['config']['blocks_per_cu'], d['config']['grid_blocks'])"
done
echo "== C5 traffic (FETCH_SIZE, WRITE_SIZE) and SQ_INSTS_VALU / SALU per trace launch, --overlap 1"
bash tools/traffic.sh "--workload c5 --no-extras --secondary none" 2>&1 | grep -v "$F" | grep "Trace" | cut -c1-200
cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_c5_valu" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --workload c5 --steps 20 --warmup 10 --no-cpu-baseline --overlap 1 --no-extras --secondary none > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"; python - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/pmc_c5_valu/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'Trace' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()): print('%-20s mean %16.1f  n %d' % (k, sum(v)/len(v), len(v)))
PY
echo "== the driver's command"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | grep -v "$F" | tail -1 > gpurun_out/r06_bench_driver.json; python - <<'PY'
import json
d = json.load(open('gpurun_out/r06_bench_driver.json'))
print({k: d.get(k) for k in ('value', 'ms_per_step', 'parity_ok', 'image_fnv', 'reference_golden')})
for k, v in d.get('secondary', {}).items():
    print(k, {q: v.get(q) for q in ('value', 'ms_per_step', 'trace_launch_ms_avg', 'parity_checked', 'parity_ok', 'image_fnv', 'frames_per_launch', 'grid_blocks')}, 'hbm', v['roofline']['frac'], 'valu', v['roofline_valu'].get('frac'))
print('cpu_baseline', d.get('cpu_baseline', {}).get('value'))
PY
echo "== the default command"
( time timeout 900 python bench.py 2>&1 | grep -v "$F" | tail -1 > gpurun_out/r06_bench_default.json ) 2>&1 | grep real; python - <<'PY'
import json
d = json.load(open('gpurun_out/r06_bench_default.json'))
print({k: d.get(k) for k in ('value', 'ms_per_step', 'parity_checked', 'image_fnv', 'reference_golden')})
for k, v in d.get('secondary', {}).items():
    print(k, {q: v.get(q) for q in ('value', 'ms_per_step', 'parity_ok')})
PY
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "config5 or time_sliced or group_bounds or per_pixel_mode" 2>&1 | grep -v "$F" | tail -5
