#!/bin/bash
# usage (on the GPU box): tools/quick_ab.sh TAG [pmc]  -- quick parity subset, the driver's command, steady state, optionally SQ counters
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
TAG=$1
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks']))"; }
echo "== [$TAG] quick parity"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "per_pixel_bit_exact or config2_1280 or overlap_is or animated or small_scenes or spp or golden or two_phase" 2>&1 | tail -3
for args in "--steps 20 --warmup 5" "--steps 200 --warmup 20" "--steps 200 --warmup 20" "--workload c3 --steps 20 --warmup 10" "--workload c5 --steps 20 --warmup 10"; do echo "-- $args"; timeout 300 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 $args 2>&1 | tail -1 | summ; done
if [ "$2" = "pmc" ]; then echo "== SQ counters at the steady-state grid"; TPT_GRID_DIV=8 bash tools/gpu_pmc.sh "--overlap 1 --no-extras" $TAG 2>&1 | grep "SQ_INSTS_VALU\|SQ_INSTS_SALU\|SQ_INSTS_BRANCH\|SQ_WAVE_CYCLES\|SQ_BUSY_CYCLES\|SQ_INSTS_VALU_TRANS\|SQ_INSTS_LDS"; fi
