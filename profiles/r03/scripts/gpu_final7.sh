#!/bin/bash
# r03: staged copies with ONE wait per call: host-path tests, bench line (no CPU leg), 1 copy thread (= round 2's copies) for comparison
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
echo "== host-path tests"; timeout 60 python -m pytest tests/test_gpu_api.py -m gpu -q --tb=short -x -k "staged or trusted or lookahead_changes" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  parity %s host %s sync %s' % (d['value'], d.get('parity_ok'), d.get('drawtest_host_ms'), d.get('sync_device_caller_ms')))"; }
echo "== bench, default copy threads"; timeout 60 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | tee gpurun_out/bench_c2_final7.json | summ
echo "== bench, TPT_HOST_COPY_THREADS=1"; TPT_HOST_COPY_THREADS=1 timeout 60 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | summ
