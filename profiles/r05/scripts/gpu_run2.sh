#!/bin/bash
# Round 5, GPU call 2: v_cndmask / integer micro-benchmarks, sweep of the tail-helper policy on the driver's command, the stagger
# experiment, the GPU suite on the shipped build (ADVICE fixes) and on the helpers build.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
O=gpurun_out/r05_2; mkdir -p $O
V=$PWD/tools/_variants/helpers
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d parity %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks'], d.get('parity_ok')))"; }
b() { timeout 200 python3 bench.py --gpus 1 --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 | summ; }
t0=$(date +%s)
echo "== ubench"; timeout 120 tools/ubench_valu > $O/ubench_valu.txt 2>&1; grep "waves/SIMD 4" $O/ubench_valu.txt | tail -20
echo "== shipped: driver's command x2"; for i in 1 2; do b --steps 20 --warmup 5; done
export TPT_LIB_DIR=$V
echo "== helpers build, helpers off x2"; for i in 1 2; do TPT_TAIL_HELPERS=0 b --steps 20 --warmup 5; done
for cfg in "1 8 2 3" "2 8 2 3" "3 8 2 3" "3 4 2 3" "3 6 3 3" "2 16 1 3" "3 8 2 10" "7 4 4 3"; do
  set -- $cfg
  echo "== helpers mult $1 max $2 newest 1/$3 pct $4"
  for i in 1 2; do TPT_HELPER_MULT=$1 TPT_HELPER_MAX=$2 TPT_HELPER_HALF=$3 TPT_HELPER_PCT=$4 b --steps 20 --warmup 5; done
  TPT_HELPER_MULT=$1 TPT_HELPER_MAX=$2 TPT_HELPER_HALF=$3 TPT_HELPER_PCT=$4 b --steps 30 --warmup 5 --parity-frames 0
done
echo "== stagger (helpers off / default helpers)"
for sg in 20 40; do
  echo "-- stagger $sg, helpers off"; for i in 1 2; do TPT_STAGGER=$sg TPT_TAIL_HELPERS=0 b --steps 20 --warmup 5; done
  TPT_STAGGER=$sg TPT_TAIL_HELPERS=0 b --steps 30 --warmup 5 --parity-frames 0
  TPT_STAGGER=$sg TPT_TAIL_HELPERS=0 b --steps 200 --warmup 20 --parity-frames 0
  echo "-- stagger $sg, helpers mult 3"; for i in 1 2; do TPT_STAGGER=$sg TPT_HELPER_MULT=3 b --steps 20 --warmup 5; done
done
echo "== helpers default: 200 steps"; b --steps 200 --warmup 20 --parity-frames 0
unset TPT_LIB_DIR
echo "elapsed $(( $(date +%s) - t0 )) s"
echo "== shipped build (ADVICE fixes): tests/test_gpu_api.py"
timeout 400 python -m pytest tests/test_gpu_api.py tests/test_bench_launcher.py -m gpu -x -q 2>&1 | grep -v "$F" | tail -6
echo "elapsed $(( $(date +%s) - t0 )) s"
echo "== helpers build: test_gpu_api.py + test_gpu_parity.py in order"
TPT_LIB_DIR=$V timeout 600 python -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -v "$F" | tail -6
echo "elapsed $(( $(date +%s) - t0 )) s"
