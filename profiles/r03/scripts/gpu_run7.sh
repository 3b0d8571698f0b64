#!/bin/bash
# r03 run 7: bench exchange test, full-line driver command (parity leg), HBM traffic per (workload, workgroups per launch)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
echo "== bench exchange agreement test"; timeout 900 python -m pytest tests/test_gpu_api.py -m gpu -q -x -k "exchanges_agree or sharded_in_process" 2>&1 | tail -5
echo "== driver's command, full line"; timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r03_bench_driver_cmd.json; python - <<'PY'
import json; d=json.load(open('gpurun_out/r03_bench_driver_cmd.json'))
print({k: d.get(k) for k in ('value','ms_per_step','exchange','rccl_ranks','image_fnv','parity_checked','parity_ok','oracle_fnv','run_rays','oracle_rays','parity_seconds')})
print('roofline', d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'][:80]); print('valu', d['roofline_valu']['frac']); print('grid', d['config']['grid_blocks'])
print({k: d.get(k) for k in ('drawtest_host_ms','sync_device_caller_ms','row_serial_Mray_s','batched_4_Mray_s','batched_8_Mray_s')}); print(d.get('cpu_baseline',{}).get('value'))
PY
for wl in c2 c3 c5; do for div in 8 4 1; do
  extra=""; [ $wl = c3 ] && extra="--steps 4 --warmup 2"; [ $wl = c5 ] && extra="--steps 4 --warmup 2"
  [ $wl != c2 ] && [ $div = 1 ] && continue
  echo "== traffic $wl griddiv $div"; TPT_GRID_DIV=$div bash tools/traffic.sh "--no-extras --workload $wl $extra" 2>&1 | grep Trace
done; done
