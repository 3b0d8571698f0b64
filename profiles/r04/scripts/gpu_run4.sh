#!/bin/bash
# r04 run 4: (run 3's kernel comparisons were void: the fixed LDS sphere area pushed the default scene 96 bytes over the launch
# code's two-workgroups-per-CU budget and every variant silently ran without LDS scene and matrix filter -- now a static_assert
# and a test.)  API + math tests, then base vs phase-2 dealing vs non-temporal stack spills, section times, C5 traversal stats.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== api + math tests"; timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_math.py -m gpu -x -q 2>&1 | grep -v "$F" | tail -6
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d bpc %d parity %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks'], d['config']['blocks_per_cu'], d.get('parity_ok')))"; }
pmc() {
  (cd /tmp && TPT_GRID_DIV=8 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d "$R/gpurun_out/r04_pmc4_$1" -o p -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --overlap 1 --no-extras --parity-frames 0 > /dev/null 2>&1)
  python - <<PY
import csv, glob, collections
for f in glob.glob('gpurun_out/r04_pmc4_$1/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'TraceQueue' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print('   '.join('%s %.1fM' % (k, sum(v)/len(v)/1e6) for k, v in sorted(acc.items())))
PY
}
for v in base deal2 deal2s3 deal1 deal3 nt; do
  if [ $v = base ]; then unset TPT_LIB; else export TPT_LIB=$R/tools/_variants/$v/libtoypathtracer_hip.so; fi
  echo "== [$v] quick parity"
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "per_pixel_bit_exact or config2_1280 or overlap_is or animated or small_scenes or spp or golden" 2>&1 | grep -v "$F" | tail -2
  echo "-- driver cmd"; timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 2>&1 | tail -1 | summ
  for i in 1 2; do echo "-- steady"; timeout 300 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --steps 200 --warmup 20 2>&1 | tail -1 | summ; done
  if [ $v = base ] || [ $v = deal2 ]; then echo "-- counters"; pmc $v; fi
done
unset TPT_LIB
echo "== section times (stats2 build)"; N=40 timeout 300 python tools/stats2_burst.py 2>&1 | grep -v "$F" | tail -14
echo "== C5 traversal stats"; timeout 300 python tools/stats_c5.py 2>&1 | grep -v "$F" | tail -5
