#!/bin/bash
# r04 run 3: full GPU suite on the tree with the short divisions / per-material constants / fixed LDS sphere area / blocking
# own stream; then A/B of kernel variants: phase-2 dealing (TPT_P2_DEAL), non-temporal stack spills
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== full GPU suite"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "$F" | tail -8
echo "== ordering test x5"; for i in 1 2 3 4 5; do timeout 300 python -m pytest tests/test_gpu_api.py -m gpu -x -q -k "default_stream_fill" 2>&1 | grep -v "$F" | tail -1; done
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d parity %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks'], d.get('parity_ok')))"; }
pmc() { # VALU / SALU / LDS instruction counts per trace launch at the steady-state grid
  (cd /tmp && TPT_GRID_DIV=8 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d "$R/gpurun_out/r04_pmc_$1" -o p -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --overlap 1 --no-extras --parity-frames 0 > /dev/null 2>&1)
  python - <<PY
import csv, glob, collections
for f in glob.glob('gpurun_out/r04_pmc_$1/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'TraceQueue' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print('   '.join('%s %.1fM' % (k, sum(v)/len(v)/1e6) for k, v in sorted(acc.items())))
PY
}
for v in base deal2 deal2s3 deal1 nt; do
  if [ $v = base ]; then unset TPT_LIB; else export TPT_LIB=$R/tools/_variants/$v/libtoypathtracer_hip.so; fi
  echo "== [$v] quick parity"
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "per_pixel_bit_exact or config2_1280 or overlap_is or animated or small_scenes or spp or golden" 2>&1 | grep -v "$F" | tail -2
  for args in "--steps 20 --warmup 5" "--steps 200 --warmup 20" "--steps 200 --warmup 20" "--workload c3 --steps 20 --warmup 10"; do echo "-- $args"; timeout 300 python bench.py --no-cpu-baseline --no-extras $args 2>&1 | tail -1 | summ; done
  echo "-- counters"; pmc $v
done
