#!/bin/bash
# r04 run 14: row-serial batches launched back to back again (rsb[1] right behind rsb[0]); the C-ABI exchange lines of C3 / C5
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
E=gpurun_out/ev4; mkdir -p $E
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
timeout 300 python -m pytest tests/test_gpu_api.py -m gpu -x -q --timeout=120 -k "seed_mode or lookahead or refused" 2>&1 | grep -v "$F" | tail -3
echo "== driver's command with the row-serial legs"; timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --extras host,row_serial 2>/dev/null | grep '^{"metric"' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('parity_ok'), 'host', d.get('drawtest_host_ms'), 'row serial', d.get('row_serial_ms'), d.get('row_serial_Mray_s'), 'batched', d.get('row_serial_batched_32_Mray_s'))"
echo "== c3 through the C-ABI exchange at N = 1 (real one-rank RCCL communicator), one frame, parity leg"; timeout 300 python bench.py --no-cpu-baseline --no-extras --workload c3 --exchange cabi --prime 0 --warmup 0 --steps 1 2>/dev/null | grep '^{"metric"' | tail -1 | tee $E/bench_c3_cabi_one_frame_parity.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['exchange'], d['rccl_ranks'], d.get('parity_ok'), d.get('image_fnv'))"
echo "== c5 through the C-ABI exchange at N = 1"; timeout 300 python bench.py --no-cpu-baseline --no-extras --workload c5 --exchange cabi --steps 20 --warmup 10 --parity-frames 0 2>/dev/null | grep '^{"metric"' | tail -1 | tee $E/bench_c5_cabi.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['exchange'], d['rccl_ranks'], d.get('image_fnv'))"
