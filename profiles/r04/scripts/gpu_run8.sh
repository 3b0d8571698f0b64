#!/bin/bash
# r04 run 8: full GPU suite after the shrink / outstanding-ticket fix; qPop snapshot A/B; non-temporal stack spills: traffic
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== full GPU suite"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "$F" | tail -8
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d bpc %d parity %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks'], d['config']['blocks_per_cu'], d.get('parity_ok')))"; }
for v in base popold base popold; do
  if [ $v = base ]; then unset TPT_LIB; else export TPT_LIB=$R/tools/_variants/$v/libtoypathtracer_hip.so; fi
  echo "== [$v]"
  echo "-- driver cmd"; timeout 300 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --steps 20 --warmup 5 2>&1 | tail -1 | summ
  echo "-- steady"; timeout 300 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --steps 200 --warmup 20 2>&1 | tail -1 | summ
done
for v in base nt; do
  if [ $v = base ]; then unset TPT_LIB; else export TPT_LIB=$R/tools/_variants/$v/libtoypathtracer_hip.so; fi
  for div in 8 4; do echo "== traffic [$v] TPT_GRID_DIV=$div"; TPT_GRID_DIV=$div bash tools/traffic.sh "--no-extras --parity-frames 0" 2>&1 | grep "TraceQueue"; done
done
