#!/bin/bash
# Round 5, GPU call 1: baseline of the shipped tree, the tail-helper build's order-dependent failure (diagnosis, not just
# reproduction), matched-pair VALU micro-benchmarks, the hardware's dynamic instruction mix by category.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
O=gpurun_out/r05_1; mkdir -p $O
V=$PWD/tools/_variants/helpers
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d valu_frac %.4f parity %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks'], d['roofline_valu']['frac'], d.get('parity_ok')))"; }
t0=$(date +%s)
echo "== baseline (shipped): driver's command x2, steady, 30, 100"
for i in 1 2; do timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | tee $O/base_driver_$i.json | summ; done
timeout 200 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --steps 200 --warmup 20 2>/dev/null | tail -1 | tee $O/base_steady.json | summ
for n in 30 100; do timeout 200 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --steps $n --warmup 5 2>/dev/null | tail -1 | summ; done
echo "elapsed $(( $(date +%s) - t0 )) s"
echo "== ubench (matched pairs)"; timeout 120 tools/ubench_valu > $O/ubench_valu.txt 2>&1; grep "waves/SIMD 4" $O/ubench_valu.txt
echo "elapsed $(( $(date +%s) - t0 )) s"
echo "== helpers build: diag (predecessor scenarios + the failing one in one process)"
TPT_LIB_DIR=$V timeout 300 python tools/helpers_diag.py --reps 6 2>&1 | grep -v "$F" | tail -30
echo "-- the same, helpers switched off at run time"
TPT_LIB_DIR=$V TPT_TAIL_HELPERS=0 timeout 300 python tools/helpers_diag.py --reps 3 2>&1 | grep -v "$F" | tail -8
echo "-- the same on the shipped library"
timeout 300 python tools/helpers_diag.py --reps 3 2>&1 | grep -v "$F" | tail -8
echo "elapsed $(( $(date +%s) - t0 )) s"
echo "== helpers build: tests/test_gpu_api.py in order, helpers on (the failing configuration of r04_run24)"
TPT_LIB_DIR=$V timeout 300 python -m pytest tests/test_gpu_api.py -m gpu -x -q 2>&1 | grep -v "$F" | tail -25
echo "elapsed $(( $(date +%s) - t0 )) s"
echo "== dynamic instruction mix by category (rocprofv3 --pmc, kernels serialised; steady-state grid 64 workgroups)"
pmc() { # name counters
  local name=$1; shift
  cd /tmp && TPT_GRID_DIV=8 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$R/gpurun_out/pmc_r05_$name" -o p -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --overlap 1 --no-extras --parity-frames 0 > /dev/null 2>&1
  cd "$R"; python - <<PY
import csv, glob, collections
for f in glob.glob('gpurun_out/pmc_r05_$name/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'TraceQueue' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()): print('%-28s mean %16.1f  n %d' % (k, sum(v)/len(v), len(v)))
PY
  rm -rf "$R/gpurun_out/pmc_r05_$name"
}
pmc mix1 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_MFMA | tee $O/pmc_mix1.txt
pmc mix2 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VSKIPPED | tee $O/pmc_mix2.txt
echo "elapsed $(( $(date +%s) - t0 )) s"
