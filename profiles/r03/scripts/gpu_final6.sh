#!/bin/bash
# r03: the tree as committed (trace-ahead first, staged copies, copy threads by host size): host-path tests, the driver's command
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
echo "== host-path tests"; timeout 120 python -m pytest tests/test_gpu_api.py -m gpu -q --tb=short -x -k "drawtest or DrawTest or trusted or lookahead" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d valu_frac %.4f parity %s host %s sync %s rowserial %s/%s batched %s/%s cpu %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks'], d['roofline_valu']['frac'], d.get('parity_ok'), d.get('drawtest_host_ms'), d.get('sync_device_caller_ms'), d.get('row_serial_Mray_s'), d.get('row_serial_batched_32_Mray_s'), d.get('batched_4_Mray_s'), d.get('batched_8_Mray_s'), d.get('cpu_baseline',{}).get('value')))"; }
echo "== driver's command (full line)"; timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | tee gpurun_out/bench_c2_driver_cmd_final6.json | summ
