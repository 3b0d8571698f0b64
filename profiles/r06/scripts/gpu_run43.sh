#!/bin/bash
# Round 6, GPU call 43: the north_star's kernel shape as a live A/B on the closing tree (tptSetKernelVariant persistent 0: one thread per
# pixel -- the lane-refill kernel with re-filling off): parity test, then configs[1] in a 100-frame stream for
#   thread per pixel + brute-force loop (the north_star's literal shape) / + packed VALU filter / + default filter,
#   lane refill (the fallback kernel), path queues (the default).
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== parity"; timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "thread_per_pixel" 2>&1 | grep "passed\|failed\|Error" | tail -3
b() { timeout 300 python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline --secondary none --parity-frames 0 "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('%8.1f Mray/s  %.4f ms/frame  %s / %s  golden %s' % (d['value'], d['ms_per_step'], d['config'].get('kernel'), d['config'].get('hit_spheres'), (d.get('reference_golden') or {}).get('ok')))"; }
echo "== thread per pixel, brute-force loop over the 46 spheres (hitSpheres 1)"; b --persistent 0 --hit-spheres 1
echo "== thread per pixel, packed VALU filter (hitSpheres 3)"; b --persistent 0 --hit-spheres 3
echo "== thread per pixel, default filter"; b --persistent 0
echo "== lane refill, brute-force loop"; b --persistent 1 --hit-spheres 1
echo "== lane refill, default filter"; b --persistent 1
echo "== path queues, packed VALU filter (no matrix cores)"; b --hit-spheres 3
echo "== path queues (the default)"; b
