#!/bin/bash
# Round 5, closing run on the final tree (GPU_MAX_HW_QUEUES=20 by default): full GPU suite, the driver's command, steady state, C5 with the
# groups' bounds on the matrix cores and on the VALU, what extra host streams cost at 20 / 32 queues.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
E=gpurun_out/final5; mkdir -p $E
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d parity %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks'], d.get('parity_ok')))"; }
t0=$(date +%s)
export TPT_ORACLE_LOG=$PWD/$E/oracle_disagreements.log
echo "== full GPU suite"; timeout 600 python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | grep "passed\|failed\|AssertionError\|^E   \|Error" | cut -c1-300 | head -30
echo "elapsed $(( $(date +%s) - t0 )) s"
echo "== checker disagreements logged:"; cat $E/oracle_disagreements.log 2>/dev/null | cut -c1-300 | head -5; echo "(end)"
echo "== driver's command (full line)"; timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | tee $E/bench_c2_driver_cmd.json | summ
echo "== driver's command again (no extras)"; timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | summ
echo "== steady state"; timeout 200 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --steps 200 --warmup 20 2>/dev/null | tail -1 | tee $E/bench_c2_steps200.json | summ
echo "== c5, bounds on the matrix cores"; timeout 200 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --workload c5 --steps 40 --warmup 20 2>/dev/null | tail -1 | tee $E/bench_c5.json | summ
echo "== c5, bounds on the VALU (--hit-spheres 3)"; for i in 1 2; do timeout 200 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --workload c5 --hit-spheres 3 --steps 40 --warmup 20 2>/dev/null | tail -1 | summ; done
echo "== 8 extra host streams"; for q in 20 32; do GPU_MAX_HW_QUEUES=$q timeout 100 python tools/extra_streams_rate.py 8 2>&1 | grep -v "$F" | tail -1; done
echo "elapsed $(( $(date +%s) - t0 )) s"
