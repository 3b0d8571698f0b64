#!/bin/bash
# r02 run 11: + sharded chunk counters, grouped claims, strided chunk order
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms pipe %.4f grid %d  host %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['pipeline_ms_per_step'], d['config']['grid_blocks'], d.get('drawtest_host_ms')))"; }
echo "== quick parity"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "per_pixel_bit_exact or config2_1280 or overlap or seventy or animated" 2>&1 | tail -3
echo "== bench"
for h in 3 0 1 8; do
for args in "--steps 20 --warmup 5" "--steps 200 --warmup 20"; do echo "-- TPT_HELP=$h $args"; TPT_HELP=$h timeout 300 python bench.py --no-cpu-baseline --no-extras $args 2>&1 | tail -1 | summ; done; done
for h in 3 0; do
cd /tmp && TPT_HELP=$h timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/burst3_h$h" -o t -- python "$R/tools/burst_trace.py" 20 > /dev/null 2>&1
cd "$R"; echo "== TPT_HELP=$h"; python tools/burst_trace.py --analyse gpurun_out/burst3_h$h | awk '{print}' | head -75
done
