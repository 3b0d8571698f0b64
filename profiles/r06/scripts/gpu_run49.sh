#!/bin/bash
# Round 6, GPU call 49: the final tree (TPT_SUPER generalised) once more: whole GPU suite, smoke, the driver's command (full line) and the default command.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  parity %s golden %s  secondary %s' % (d['value'], d['ms_per_step'], d.get('parity_ok'), (d.get('reference_golden') or {}).get('ok'), {k: (round(v['value'], 1), v.get('parity_ok')) for k, v in (d.get('secondary') or {}).items()}))"; }
echo "== full GPU suite"; timeout 1700 python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | grep "passed\|failed\|AssertionError\|^E   \|Error\|oracle self-check" | cut -c1-300 | head -20
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v "$F" | tail -2
echo "== driver's command"; timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | tee gpurun_out/r06_bench_driver_final2.json | summ
echo "== default command"; timeout 600 python bench.py 2>/dev/null | tail -1 | tee gpurun_out/r06_bench_default_final2.json | summ
