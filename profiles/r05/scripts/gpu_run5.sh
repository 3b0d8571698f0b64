#!/bin/bash
# Round 5, GPU call 5: self-cull A/B (shipped tree = cull with the record held in registers; cull2 = recomputed at the use; nocull =
# round-4 kernel), then the helpers build's whole test_gpu_api.py five more times with the failing images kept for analysis.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
O=gpurun_out/r05_5; mkdir -p $O
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d parity %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks'], d.get('parity_ok')))"; }
b() { timeout 200 python3 bench.py --gpus 1 --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 | summ; }
t0=$(date +%s)
for v in "" nocull cull2 "" nocull cull2; do
  if [ -n "$v" ]; then export TPT_LIB_DIR=$PWD/tools/_variants/$v; else unset TPT_LIB_DIR; fi
  echo "== [${v:-shipped(cull1)}] driver's command, steady x2, c3"
  b --steps 20 --warmup 5
  b --steps 200 --warmup 20 --parity-frames 0
  b --steps 200 --warmup 20 --parity-frames 0
  b --workload c3 --steps 20 --warmup 10 --parity-frames 0
done
unset TPT_LIB_DIR
echo "== quick parity on the shipped tree (cull1) and cull2"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "per_pixel_bit_exact or config2_1280 or overlap_is or animated or small_scenes or spp or golden or two_phase" 2>&1 | grep -v "$F" | tail -3
TPT_LIB_DIR=$PWD/tools/_variants/cull2 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "per_pixel_bit_exact or config2_1280 or overlap_is or animated or small_scenes or spp or golden or two_phase" 2>&1 | grep -v "$F" | tail -3
echo "elapsed $(( $(date +%s) - t0 )) s"
V=$PWD/tools/_variants/helpers
for i in 1 2 3 4 5; do
  echo "== helpers build, run $i"; TPT_MISMATCH_DUMP=$PWD/gpurun_out/r05_5/dump TPT_LIB_DIR=$V timeout 300 python -m pytest tests/test_gpu_api.py -m gpu -q 2>&1 | grep -v "$F" | grep "passed\|failed\|AssertionError\|E    \|saved" | cut -c1-300
done
echo "elapsed $(( $(date +%s) - t0 )) s"
