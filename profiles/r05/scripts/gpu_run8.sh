#!/bin/bash
# Round 5, GPU call 8: full GPU suite on the tree with the queue-setup fix of the test process and the de-hoisted scalars (SGPR spills
# 63 -> 33); A/B of that kernel on the driver's command / steady state / C3; C5 with the 128-register grouped instantiation against
# the 120-register one; HBM traffic and executed VALU instructions per launch (C2, C5).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
O=gpurun_out/r05_8; mkdir -p $O
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d parity %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks'], d.get('parity_ok')))"; }
b() { timeout 200 python3 bench.py --gpus 1 --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 | summ; }
t0=$(date +%s)
export TPT_ORACLE_LOG=$PWD/$O/oracle_disagreements.log TPT_MISMATCH_DUMP=$PWD/$O/dump
echo "== full GPU suite"; timeout 600 python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | grep "passed\|failed\|AssertionError\|^E   \|saved\|Error" | cut -c1-300 | head -30
echo "== checker disagreements logged:"; cat $O/oracle_disagreements.log 2>/dev/null | cut -c1-300 | head -10; echo "(end)"
echo "elapsed $(( $(date +%s) - t0 )) s"
echo "== driver's command x3, steady x2, c3"; for i in 1 2 3; do b --steps 20 --warmup 5; done; for i in 1 2; do b --steps 200 --warmup 20 --parity-frames 0; done; b --workload c3 --steps 20 --warmup 10 --parity-frames 0
echo "== c5: product (128 registers for the grouped instantiation) x2"; for i in 1 2; do b --workload c5 --steps 20 --warmup 10 --parity-frames 0; done
echo "== c5: 120 registers x2"; for i in 1 2; do TPT_LIB_DIR=$PWD/tools/_variants/c5v120 b --workload c5 --steps 20 --warmup 10 --parity-frames 0; done
echo "elapsed $(( $(date +%s) - t0 )) s"
pmc() { # name, bench args, counters...
  local name=$1; local args=$2; shift; shift
  cd /tmp && TPT_GRID_DIV=8 timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$R/gpurun_out/pmc_r05_$name" -o p -- python "$R/bench.py" $args --no-cpu-baseline --overlap 1 --no-extras --parity-frames 0 > /dev/null 2>&1
  cd "$R"; python - <<PY
import csv, glob, collections
for f in glob.glob('gpurun_out/pmc_r05_$name/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'TraceQueue' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()): print('$name %-24s mean %16.1f  n %d' % (k, sum(v)/len(v), len(v)))
PY
  rm -rf "$R/gpurun_out/pmc_r05_$name"
}
echo "== C2 at 64 workgroups per launch: FETCH_SIZE / WRITE_SIZE (separate passes), SQ_INSTS_VALU"
pmc c2f "--steps 10 --warmup 2" FETCH_SIZE | tee $O/pmc_c2.txt
pmc c2w "--steps 10 --warmup 2" WRITE_SIZE | tee -a $O/pmc_c2.txt
pmc c2v "--steps 10 --warmup 2" SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU | tee -a $O/pmc_c2.txt
echo "== C5 at 64 workgroups per launch"
pmc c5f "--workload c5 --steps 3 --warmup 1" FETCH_SIZE | tee $O/pmc_c5.txt
pmc c5w "--workload c5 --steps 3 --warmup 1" WRITE_SIZE | tee -a $O/pmc_c5.txt
pmc c5v "--workload c5 --steps 3 --warmup 1" SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES | tee -a $O/pmc_c5.txt
echo "elapsed $(( $(date +%s) - t0 )) s"
