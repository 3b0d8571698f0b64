#!/bin/bash
# r03 closing run after the host-path changes (staged copies, trace-ahead launched beside the DMA): full GPU suite, smoke, the driver's command
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
echo "== full GPU suite"; timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -25
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d valu_frac %.4f parity %s host %s sync %s rowserial %s/%s batched %s/%s cpu %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks'], d['roofline_valu']['frac'], d.get('parity_ok'), d.get('drawtest_host_ms'), d.get('sync_device_caller_ms'), d.get('row_serial_Mray_s'), d.get('row_serial_batched_32_Mray_s'), d.get('batched_4_Mray_s'), d.get('batched_8_Mray_s'), d.get('cpu_baseline',{}).get('value')))"; }
echo "== driver's command (full line)"; timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | tee gpurun_out/bench_c2_driver_cmd_final5.json | summ
