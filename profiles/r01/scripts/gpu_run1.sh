#!/bin/bash
# First GPU session: smoke, parity tests, bench + variants, rocprof kernel trace, VALU microbench.
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
echo "== rocminfo"; rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8
echo "== nproc $(nproc)"; lscpu | grep -E "Model name|^CPU\(s\)" 
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25
echo "== bench default"; timeout 600 python bench.py --steps 100 --warmup 10 2>&1 | tail -3 | tee gpurun_out/bench_c2.json
echo "== variants"
for v in "--fold 1" "--hit-spheres 1" "--persistent 0" "--persistent 0 --hit-spheres 1" "--lds-scene 0"; do
  echo "-- $v"; timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline $v 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms_per_step'], d['kernel_Mray_s'], d['config']['blocks_per_cu'], d['config']['grid_blocks'])"
done
echo "== c3"; timeout 600 python bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_c3.json
echo "== c5"; timeout 600 python bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_c5.json
echo "== ubench"; timeout 300 ./tools/ubench_valu 2>&1 | tee gpurun_out/ubench_valu.txt
echo "== rocprof kernel trace"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_c2" -o c2 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -3
cd "$GRAFT_REPO_ROOT"; find gpurun_out/prof_c2 -name "*stats*" | head; for f in $(find gpurun_out/prof_c2 -name "*kernel_stats.csv"); do head -5 $f; done
