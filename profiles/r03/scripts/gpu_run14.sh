#!/bin/bash
# r03 run 14: pruned tree + row-serial batches + bounded slot memory: full GPU suite, the driver's command (full line)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
echo "== full GPU suite"; timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8
echo "== driver's command, full line"; timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r03_bench_driver_cmd.json; python - <<'PY'
import json; d=json.load(open('gpurun_out/r03_bench_driver_cmd.json'))
print({k: d.get(k) for k in ('value','ms_per_step','exchange','image_fnv','parity_checked','parity_ok','run_rays','oracle_rays')})
print('roofline', d['roofline']['frac'], d['roofline']['traffic']); print('valu', d['roofline_valu']['frac']); print('grid', d['config']['grid_blocks'])
print({k: d.get(k) for k in ('drawtest_host_ms','sync_device_caller_ms','row_serial_Mray_s','row_serial_batched_32_Mray_s','batched_4_Mray_s','batched_8_Mray_s')}); print(d.get('cpu_baseline',{}).get('value'))
PY
