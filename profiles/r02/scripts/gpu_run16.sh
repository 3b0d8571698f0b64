#!/bin/bash
# r02 run 16: frame table in device memory: quota sweep
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms pipe %.4f grid %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['pipeline_ms_per_step'], d['config']['grid_blocks']))"; }
for h in 0 6 12 32 1000; do
for args in "--steps 20 --warmup 5" "--steps 200 --warmup 20"; do echo "-- TPT_HELP=$h $args"; TPT_TABLE=device TPT_HELP=$h timeout 300 python bench.py --no-cpu-baseline --no-extras $args 2>&1 | tail -1 | summ; done; done
for h in 1000; do
cd /tmp && TPT_TABLE=device TPT_HELP=$h timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/burst6_h$h" -o t -- python "$R/tools/burst_trace.py" 20 > /dev/null 2>&1
cd "$R"; echo "== TPT_TABLE=device TPT_HELP=$h"; python tools/burst_trace.py --analyse gpurun_out/burst6_h$h | head -70
done
