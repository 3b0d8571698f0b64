#!/bin/bash
# Round 5, GPU call 25: call 24 put the rare difference on the matrix-core filter of the groups' bounds (A tiles read from GLOBAL memory;
# the same filter with its table in LDS never showed it).  The compiler gives the second MFMA of a pair a destination that overlaps its own
# A operand (v[16:31] <- A v[16:19]) and issues it while the next A tile's load is still in flight.  GPU_MAX_HW_QUEUES=32 + 16 extra streams:
# (1) both A tiles loaded and arrived before the first MFMA (overlap kept); (2) A registers kept live across the MFMAs (no overlap).
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=32
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step' % (d['value'], d['ms_per_step']))"; }
run() { echo "== $1"; n=$2; shift; shift; env "$@" C5_PATH=device C5_DISTURB=torch_streams timeout 400 python tools/c5_after_hooks.py $n keep 2>&1 | grep -v "$F" | grep "results\|rror" | tail -3; }
run "1: loads arrived before the MFMAs" 50 TPT_LIB_DIR=$PWD/tools/_variants/mxwait
run "2: no destination / A overlap" 50 TPT_LIB_DIR=$PWD/tools/_variants/mxkeep
export GPU_MAX_HW_QUEUES=20
for v in mxwait mxkeep; do echo "== C5 rate: $v"; for i in 1 2; do TPT_LIB_DIR=$PWD/tools/_variants/$v timeout 200 python3 bench.py --gpus 1 --no-cpu-baseline --no-extras --workload c5 --steps 20 --warmup 10 --parity-frames 0 2>/dev/null | tail -1 | summ; done; done
