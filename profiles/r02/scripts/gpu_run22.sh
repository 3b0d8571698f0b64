#!/bin/bash
# r02 run 22: back to one launch = one frame (frame table removed), raised priority for the resolve kernels, queue probe
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms pipe %.4f grid %d  host %s  rowserial %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['pipeline_ms_per_step'], d['config']['grid_blocks'], d.get('drawtest_host_ms'), d.get('row_serial_Mray_s')))"; }
for args in "--steps 20 --warmup 5" "--steps 20 --warmup 5" "--steps 200 --warmup 20" "--workload c3 --steps 20 --warmup 10" "--workload c5 --steps 20 --warmup 10" "--animate" "--overlap 1 --steps 50" "--overlap 2 --steps 50"; do echo "-- $args"; timeout 300 python bench.py --no-cpu-baseline $args 2>&1 | tail -1 | summ; done
echo "== stats2"; timeout 60 python tools/stats2_burst.py 2>&1 | grep -v amdgpu.ids
echo "== full gpu suite"
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8
echo "== full bench line (driver's command)"
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | tail -1
