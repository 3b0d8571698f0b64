#!/bin/bash
# Round 5, GPU call 7: the new default build (tail helpers adopted, ADVICE fixes, knobs pruned): the full GPU suite twice (checker in
# redundant mode, disagreements logged), then the driver's command x3 and 30 / 100 / 200 frames.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
O=gpurun_out/r05_7; mkdir -p $O
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d valu_frac %.4f parity %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks'], d['roofline_valu']['frac'], d.get('parity_ok')))"; }
b() { timeout 200 python3 bench.py --gpus 1 --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 | summ; }
t0=$(date +%s)
export TPT_ORACLE_LOG=$PWD/$O/oracle_disagreements.log TPT_MISMATCH_DUMP=$PWD/$O/dump
for i in 1 2; do
  echo "== full GPU suite, run $i"; timeout 600 python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | grep "passed\|failed\|AssertionError\|^E   \|saved\|Error" | cut -c1-300 | head -40
  echo "elapsed $(( $(date +%s) - t0 )) s"
done
echo "== checker disagreements logged:"; cat $O/oracle_disagreements.log 2>/dev/null | cut -c1-300 | head -20; echo "(end)"
echo "== driver's command x3"; for i in 1 2 3; do b --steps 20 --warmup 5; done
echo "== 30 / 100 / 200 frames"; b --steps 30 --warmup 5 --parity-frames 0; b --steps 100 --warmup 5 --parity-frames 0; b --steps 200 --warmup 20 --parity-frames 0
echo "== helpers off (TPT_TAIL_HELPERS=0): driver's command x2"; for i in 1 2; do TPT_TAIL_HELPERS=0 b --steps 20 --warmup 5; done
echo "elapsed $(( $(date +%s) - t0 )) s"
