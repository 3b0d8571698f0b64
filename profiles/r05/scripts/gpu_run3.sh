#!/bin/bash
# Round 5, GPU call 3: paired v_cndmask micro-benchmarks; tests/test_gpu_api.py in order (subprocess tests deselected), four times
# on the helpers build and three times on the shipped build, with the mismatch diagnostics of the sharded tests.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
O=gpurun_out/r05_3; mkdir -p $O
V=$PWD/tools/_variants/helpers
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
t0=$(date +%s)
echo "== ubench"; timeout 120 tools/ubench_valu > $O/ubench_valu.txt 2>&1; grep "waves/SIMD 4" $O/ubench_valu.txt | tail -6
K="not cxx and not bench and not initialised_hip_first"
for i in 1 2 3 4; do
  echo "== helpers build, run $i"; TPT_LIB_DIR=$V timeout 300 python -m pytest tests/test_gpu_api.py -m gpu -q -k "$K" 2>&1 | grep -v "$F" | grep -v "^$" | tail -40 | cut -c1-260
done
for i in 1 2 3; do
  echo "== shipped build, run $i"; timeout 300 python -m pytest tests/test_gpu_api.py -m gpu -q -k "$K" 2>&1 | grep -v "$F" | grep -v "^$" | tail -12 | cut -c1-260
done
echo "elapsed $(( $(date +%s) - t0 )) s"
