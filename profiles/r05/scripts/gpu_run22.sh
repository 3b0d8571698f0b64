#!/bin/bash
# Round 5, GPU call 22: the grouped kernel with 64 lights asks for exactly 81 920 B of LDS: two workgroups fill a CU's 160 KB to the last
# byte.  Is THAT what the rare difference under queue time-slicing needs?  (a) 1 280 B more (one workgroup per CU), (b) pair lists of 128
# entries instead of 192 (79 872 B: two workgroups and 4 KB to spare).  16 extra streams in the process; then each build's C5 rate.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks']))"; }
run() { echo "== $1"; n=$2; shift; shift; env "$@" C5_PATH=device C5_DISTURB=torch_streams timeout 400 python tools/c5_after_hooks.py $n keep 2>&1 | grep -v "$F" | grep "results\|rror" | tail -3; }
rate() { for i in 1 2; do env "$@" timeout 200 python3 bench.py --gpus 1 --no-cpu-baseline --no-extras --workload c5 --steps 20 --warmup 10 --parity-frames 0 2>/dev/null | tail -1 | summ; done; }
run "a: +1280 B of LDS" 40 TPT_LIB_DIR=$PWD/tools/_variants/ldspad
run "b: pair lists of 128 entries" 40 TPT_LIB_DIR=$PWD/tools/_variants/cap128
echo "== C5 rate: a"; rate TPT_LIB_DIR=$PWD/tools/_variants/ldspad
echo "== C5 rate: b"; rate TPT_LIB_DIR=$PWD/tools/_variants/cap128
echo "== C5 rate: shipped"; rate
