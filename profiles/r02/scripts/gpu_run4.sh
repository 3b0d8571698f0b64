#!/bin/bash
# r02 run 4: matrix filter A/B with 16 slots again (32 trace streams oversubscribed the 32 hardware queues)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms pipe %.4f grid %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['pipeline_ms_per_step'], d['config']['grid_blocks']))"; }
echo "== bench A/B"
for args in "--steps 200 --warmup 20" "--steps 200 --warmup 20 --hit-spheres 3" "--steps 20 --warmup 5" "--steps 20 --warmup 5 --hit-spheres 3" "--workload c3 --steps 20 --warmup 10" "--workload c3 --steps 20 --warmup 10 --hit-spheres 3" "--overlap 1 --steps 50" "--overlap 1 --steps 50 --hit-spheres 3"; do echo "-- $args"; timeout 300 python bench.py --no-cpu-baseline $args 2>&1 | tail -1 | summ; done
echo "-- vgpr128 variant"; TPT_LIB=$R/tools/_variants/vgpr128/libtoypathtracer_hip.so timeout 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>&1 | tail -1 | summ
echo "== sharded exchange test x5"
for i in 1 2 3 4 5; do timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sharded_frame_exchange" 2>&1 | tail -3; done
echo "== full gpu suite"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8
