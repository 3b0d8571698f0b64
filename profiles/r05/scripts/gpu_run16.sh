#!/bin/bash
# Round 5, GPU call 16: is it the scratch memory?  The grouped kernel built without a single spilled register (166 VGPRs, no private
# segment) under the tool that shows the rare C5 difference; and what that build costs.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks']))"; }
export TPT_LIB_DIR=$PWD/tools/_variants/g96w2
echo "== no-scratch grouped kernel, hooks context kept alive, 200 renders x2"
for i in 1 2; do timeout 300 python tools/c5_after_hooks.py 100 keep 2>&1 | grep -v "$F" | grep "results\|rror" | tail -3; done
echo "== its C5 rate"; for i in 1 2; do timeout 200 python3 bench.py --gpus 1 --no-cpu-baseline --no-extras --workload c5 --steps 20 --warmup 10 --parity-frames 0 2>/dev/null | tail -1 | summ; done
unset TPT_LIB_DIR
