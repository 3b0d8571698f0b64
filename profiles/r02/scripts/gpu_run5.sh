#!/bin/bash
# r02 run 5: frame table (workers go on with newer frames): correctness + burst / steady-state rates, vs TPT_HELP=0
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms pipe %.4f grid %d  host %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['pipeline_ms_per_step'], d['config']['grid_blocks'], d.get('drawtest_host_ms')))"; }
echo "== quick parity"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "per_pixel_bit_exact or config2 or overlap or seventy or animated" 2>&1 | tail -4
echo "== bench"
for h in 3 0 1 6; do
for args in "--steps 20 --warmup 5" "--steps 200 --warmup 20"; do echo "-- TPT_HELP=$h $args"; TPT_HELP=$h timeout 300 python bench.py --no-cpu-baseline --no-extras $args 2>&1 | tail -1 | summ; done; done
echo "-- default x2, steps 20"; for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>&1 | tail -1 | summ; done
echo "-- c3"; timeout 300 python bench.py --no-cpu-baseline --no-extras --workload c3 --steps 20 --warmup 10 2>&1 | tail -1 | summ
echo "-- c5"; timeout 300 python bench.py --no-cpu-baseline --no-extras --workload c5 --steps 20 --warmup 10 2>&1 | tail -1 | summ
echo "-- matrix variant (compile-time)"; TPT_LIB=$R/tools/_variants/matrix/libtoypathtracer_hip.so timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 20 2>&1 | tail -1 | summ
echo "-- animate"; timeout 300 python bench.py --no-cpu-baseline --no-extras --animate 2>&1 | tail -1 | summ
echo "== full gpu suite"
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8
echo "== full bench line (driver's command)"
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | tail -1
