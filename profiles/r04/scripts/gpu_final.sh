#!/bin/bash
# r04 closing run on the final tree: full GPU suite (default = stream batching on), again with stream batching off, smoke(), and
# the C5 evidence re-taken with 8 members per group (bench line, rocprofv3 kernel stats, traffic, SQ counters)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
E=gpurun_out/ev4; mkdir -p $E
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== full GPU suite"; timeout 600 python -m pytest tests -m gpu -x -q --timeout=200 2>&1 | grep -v "$F" | tail -5
echo "== full GPU suite, TPT_FORCE_STREAM_BATCH=0"; TPT_FORCE_STREAM_BATCH=0 timeout 600 python -m pytest tests -m gpu -x -q --timeout=200 2>&1 | grep -v "$F" | tail -4
echo "== smoke"; timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" | tail -2
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d valu_frac %.4f' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks'], d['roofline_valu']['frac']))"; }
echo "== c5"; timeout 200 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --workload c5 --steps 40 --warmup 20 2>/dev/null | grep '^{"metric"' | tail -1 | tee $E/bench_c5.json | summ
echo "== c5 through the C-ABI exchange at N = 1"; timeout 300 python bench.py --no-cpu-baseline --no-extras --workload c5 --exchange cabi --steps 20 --warmup 10 --parity-frames 0 2>/dev/null | grep '^{"metric"' | tail -1 | tee $E/bench_c5_cabi.json | summ
echo "== rocprofv3 kernel stats, c5"; cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$E/prof_c5" -o k -- python3 "$R/bench.py" --workload c5 --steps 20 --warmup 10 --no-cpu-baseline --no-extras --parity-frames 0 > "$R/$E/prof_c5_bench.json" 2>/dev/null; cd "$R"; cp $E/prof_c5/k_kernel_stats.csv $E/prof_c5_kernel_stats.csv; head -3 $E/prof_c5_kernel_stats.csv | cut -c1-200; rm -rf $E/prof_c5
echo "== traffic c5 griddiv 8"; TPT_GRID_DIV=8 bash tools/traffic.sh "--no-extras --parity-frames 0 --workload c5 --steps 4 --warmup 2" 2>&1 | grep Trace
echo "== SQ counters c5"; cd /tmp && TPT_GRID_DIV=8 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM --output-format csv -d "$R/gpurun_out/pmc_r04c5b" -o p -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --overlap 1 --no-extras --parity-frames 0 --workload c5 > /dev/null 2>&1; cd "$R"; python - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/pmc_r04c5b/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'Trace' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()): print('%-28s mean %16.1f  n %d' % (k, sum(v)/len(v), len(v)))
PY
