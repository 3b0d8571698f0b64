#!/bin/bash
# r02 run 14: frame table in device memory (DMA-published) vs pinned host memory
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms pipe %.4f grid %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['pipeline_ms_per_step'], d['config']['grid_blocks']))"; }
for t in device host; do for h in 3 1; do
for args in "--steps 20 --warmup 5" "--steps 200 --warmup 20"; do echo "-- TPT_TABLE=$t TPT_HELP=$h $args"; TPT_TABLE=$t TPT_HELP=$h timeout 300 python bench.py --no-cpu-baseline --no-extras $args 2>&1 | tail -1 | summ; done; done; done
for t in device; do for h in 3; do echo "== stats2 TPT_TABLE=$t TPT_HELP=$h"; TPT_TABLE=$t TPT_HELP=$h timeout 40 python tools/stats2_burst.py 2>&1 | grep -v amdgpu.ids; done; done
