#!/bin/bash
# r03 run 11: matrix-core filter (f16-split MFMA) as the default: full GPU suite, benches, SQ counters
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
echo "== full GPU suite"; timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d parity %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks'], d.get('parity_ok')))"; }
for args in "--steps 20 --warmup 5" "--steps 200 --warmup 20" "--workload c3 --steps 20 --warmup 10" "--workload c5 --steps 20 --warmup 10" "--workload c1 --steps 200 --warmup 20" "--hit-spheres 3 --steps 200 --warmup 20"; do echo "-- $args"; timeout 300 python bench.py --no-cpu-baseline --no-extras $args 2>&1 | tail -1 | summ; done
echo "== SQ counters at the steady-state grid"; TPT_GRID_DIV=8 bash tools/gpu_pmc.sh "--overlap 1 --no-extras" mx 2>&1 | grep "SQ_\|GRBM"
