#!/bin/bash
# Round-2 closing run on the committed tree: smoke(), the whole GPU suite, the driver's command (full JSON line) and the steady state.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
mkdir -p gpurun_out/ev4
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d host %.3f sync %.3f rowserial %.0f batched4 %.0f batched8 %.0f cpu %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks'], d.get('drawtest_host_ms'), d.get('sync_device_caller_ms'), d.get('row_serial_Mray_s'), d.get('batched_4_Mray_s'), d.get('batched_8_Mray_s'), (d.get('cpu_baseline') or {}).get('value')))"; }
echo "== driver's command (full line)"; timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | tee gpurun_out/ev4/bench_c2_driver_cmd.json | summ
echo "== steady state"; timeout 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | tail -1 | tee gpurun_out/ev4/bench_c2_steps200.json | summ
