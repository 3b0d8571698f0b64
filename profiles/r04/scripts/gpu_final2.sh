#!/bin/bash
# r04 closing run 2 (after the last source change: tdivByPi lets +0 take the short form, as its comment says): full GPU suite, smoke, the driver's command
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== full GPU suite"; timeout 600 python -m pytest tests -m gpu -x -q --timeout=200 2>&1 | grep -v "$F" | tail -4
echo "== smoke"; timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" | tail -1
echo "== driver's command"; timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{"metric"' | tail -1 | tee gpurun_out/ev4/bench_c2_driver_cmd_final.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('parity_ok'), d['roofline']['frac'], d['roofline']['traffic'], d['roofline_valu']['frac'], d['cpu_baseline']['value'], d.get('drawtest_host_ms'), d.get('row_serial_Mray_s'), d.get('row_serial_batched_32_Mray_s'))"
echo "== steady"; timeout 100 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --steps 200 --warmup 20 2>/dev/null | grep '^{"metric"' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
