#!/bin/bash
# r04 run 12: full GPU suite on the committed tree (5d2d53d: qPop fix, LDS-typed rings, dealt grouped traversal with matrix-core
# bounds and dealt exact tests), C2 / C5 rates, C5 traversal stats.  Every step under a short timeout.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== full GPU suite"; timeout 600 python -m pytest tests -m gpu -x -q --timeout=200 2>&1 | grep -v "$F" | tail -8
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d bpc %d parity %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks'], d['config']['blocks_per_cu'], d.get('parity_ok')))"; }
echo "== C2 driver cmd / steady x2"
timeout 120 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 2>&1 | tail -1 | summ
for i in 1 2; do timeout 120 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --steps 200 --warmup 20 2>&1 | tail -1 | summ; done
for v in base gde0 base; do
  if [ $v = base ]; then unset TPT_LIB; else export TPT_LIB=$R/tools/_variants/$v/libtoypathtracer_hip.so; fi
  echo "== [$v] c5"; timeout 120 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --workload c5 --steps 20 --warmup 10 2>&1 | tail -1 | summ
done
unset TPT_LIB
echo "== C5 traversal stats"; timeout 90 python tools/stats_c5.py 2>&1 | grep -v "$F" | tail -4
echo "== section times (stats2 build)"; N=40 timeout 90 python tools/stats2_burst.py 2>&1 | grep -v "$F" | tail -6
