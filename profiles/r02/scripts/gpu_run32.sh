#!/bin/bash
# r02 run 32: Config.h switches at run time, overlap stress test, speed check
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms pipe %.4f grid %d host %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['pipeline_ms_per_step'], d['config']['grid_blocks'], d.get('drawtest_host_ms')))"; }
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "config_switches or stress_300" 2>&1 | tail -6 | cut -c1-300
for args in "--steps 200 --warmup 20" "--steps 20 --warmup 5"; do echo "-- $args"; timeout 300 python bench.py --no-cpu-baseline $args 2>&1 | tail -1 | summ; done
