#!/bin/bash
# r02 run 20: frame table published by a 1-wave kernel with raised priority
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms pipe %.4f grid %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['pipeline_ms_per_step'], d['config']['grid_blocks']))"; }
for h in 0 3 12 1000; do
for args in "--steps 20 --warmup 5" "--steps 200 --warmup 20"; do echo "-- TPT_HELP=$h $args"; TPT_HELP=$h timeout 300 python bench.py --no-cpu-baseline --no-extras $args 2>&1 | tail -1 | summ; done; done
echo "== stats2 HELP=12"; TPT_HELP=12 timeout 40 python tools/stats2_burst.py 2>&1 | grep -v amdgpu.ids
for h in 12; do
cd /tmp && TPT_HELP=$h timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/burst9_h$h" -o t -- python "$R/tools/burst_trace.py" 20 > /dev/null 2>&1
cd "$R"; echo "== TPT_HELP=$h"; python tools/burst_trace.py --analyse gpurun_out/burst9_h$h | head -64
done
