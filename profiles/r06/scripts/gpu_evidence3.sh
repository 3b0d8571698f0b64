#!/bin/bash
# Round 6, THIRD evidence run (closing tree: three-stage dealing with the half-line test on the bounds, deferred sharded batches): full GPU suite, the driver's command (full line incl. the secondary legs and the CPU baseline),
# the default command, C3 / C5 / C1 lines, rocprofv3 --kernel-trace --stats of the driver's command and of the steady stream, HBM traffic
# and executed VALU instructions per launch for C2 and C5 at the stream's launch geometry (separate --pmc passes, TPT_GRID_DIV=8, --overlap 1),
# the loopback table, the time-sliced child.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
E=gpurun_out/evidence8; rm -rf $E; mkdir -p $E
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d parity %s golden %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks'], d.get('parity_ok'), (d.get('reference_golden') or {}).get('ok')))"; }
t0=$(date +%s)
export TPT_ORACLE_LOG=$PWD/$E/oracle_disagreements.log
echo "== full GPU suite"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | grep "passed\|failed\|AssertionError\|^E   \|Error\|oracle self-check" | cut -c1-300 | head -30
echo "elapsed $(( $(date +%s) - t0 )) s"
echo "== checker disagreements logged:"; cat $E/oracle_disagreements.log 2>/dev/null | cut -c1-300 | head -5; echo "(end)"
echo "== driver's command (full line)"; timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | tee $E/bench_c2_driver_cmd.json | summ
python - <<PY
import json
d = json.load(open('$E/bench_c2_driver_cmd.json'))
for k, v in d.get('secondary', {}).items():
    print('   secondary', k, '%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d parity %s  hbm frac %.5f  valu frac %s' % (v['value'], v['ms_per_step'], v['trace_launch_ms_avg'], v['grid_blocks'], v.get('parity_ok'), v['roofline']['frac'], v['roofline_valu'].get('frac')))
print('   cpu_baseline', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], {k: round(v['value'], 1) for k, v in d['cpu_baseline'].get('builds', {}).items()})
print('   roofline', {k: d['roofline'][k] for k in ('achieved', 'frac', 'traffic')}, 'valu', {k: d['roofline_valu'][k] for k in ('achieved', 'frac')}, 'host path', d.get('drawtest_host_ms'), 'row serial', d.get('row_serial_Mray_s'), d.get('row_serial_batched_32_Mray_s'))
PY
echo "== driver's command again (no extras)"; for i in 1 2; do timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --secondary none 2>/dev/null | tail -1 | summ; done
echo "== the default command"; ( time timeout 600 python bench.py 2>/dev/null | tail -1 > $E/bench_c2_default.json ) 2>&1 | grep real; summ < $E/bench_c2_default.json
echo "== 30 / 100 frames"; for k in 30 100; do timeout 200 python bench.py --no-cpu-baseline --no-extras --secondary none --parity-frames 0 --steps $k --warmup 5 2>/dev/null | tail -1 | summ; done
echo "== c3 steady"; timeout 300 python bench.py --no-cpu-baseline --no-extras --secondary none --parity-frames 0 --workload c3 --steps 40 --warmup 10 2>/dev/null | tail -1 | tee $E/bench_c3.json | summ
echo "== c5"; for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-extras --secondary none --workload c5 --steps 40 --warmup 20 2>/dev/null | tail -1 | tee $E/bench_c5.json | summ; done
echo "== c5, flat VALU filter (--hit-spheres 3)"; timeout 300 python bench.py --no-cpu-baseline --no-extras --secondary none --workload c5 --hit-spheres 3 --steps 40 --warmup 20 2>/dev/null | tail -1 | summ
echo "== c5, bounds on the matrix cores (hooks build, --hit-spheres 4)"; TPT_LIB=$R/toypathtracer_amd/lib/libtoypathtracer_hip_hooks.so timeout 300 python bench.py --no-cpu-baseline --no-extras --secondary none --workload c5 --hit-spheres 4 --steps 40 --warmup 20 2>/dev/null | tail -1 | summ
echo "== c1"; timeout 200 python bench.py --no-cpu-baseline --no-extras --secondary none --workload c1 --steps 400 --warmup 40 --parity-frames 0 2>/dev/null | tail -1 | tee $E/bench_c1.json | summ
echo "elapsed $(( $(date +%s) - t0 )) s"
echo "== rocprofv3 --kernel-trace --stats: the driver's command"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$E/prof_driver -o p -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --secondary none > /dev/null 2>&1; cd $R
find $E/prof_driver -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $E/prof_driver_cmd_kernel_stats.csv; cat $E/prof_driver_cmd_kernel_stats.csv | cut -c1-200 | head -8
echo "== rocprofv3 --kernel-trace --stats: 200-frame stream"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$E/prof_stream -o p -- python3 $R/bench.py --no-cpu-baseline --no-extras --secondary none --parity-frames 0 > /dev/null 2>&1; cd $R
find $E/prof_stream -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $E/prof_stream_kernel_stats.csv; cat $E/prof_stream_kernel_stats.csv | cut -c1-200 | head -6
echo "== rocprofv3 --kernel-trace --stats: c5, 40 frames"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$E/prof_c5 -o p -- python3 $R/bench.py --no-cpu-baseline --no-extras --secondary none --workload c5 --steps 40 --warmup 20 > /dev/null 2>&1; cd $R
find $E/prof_c5 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $E/prof_c5_kernel_stats.csv; cat $E/prof_c5_kernel_stats.csv | cut -c1-200 | head -6
rm -rf $E/prof_driver $E/prof_stream $E/prof_c5
pmc() { # name, bench args, counters...
  local name=$1 args=$2; shift; shift
  cd /tmp && TPT_GRID_DIV=8 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$R/gpurun_out/pmc_r06_$name" -o p -- python "$R/bench.py" $args --no-cpu-baseline --overlap 1 --no-extras --secondary none --parity-frames 0 > /dev/null 2>&1
  cd "$R"; python - <<PY
import csv, glob, collections
for f in glob.glob('gpurun_out/pmc_r06_$name/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'Trace' in r['Kernel_Name']: acc[(r['Kernel_Name'][:44], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()): print('%-46s %-18s mean %18.1f  n %d' % (k[0], k[1], sum(v)/len(v), len(v)))
PY
  rm -rf "$R/gpurun_out/pmc_r06_$name"
}
echo "== PMC (separate passes; per trace launch; 64 workgroups per launch)"
pmc c2f "--steps 10 --warmup 2" FETCH_SIZE | tee $E/pmc_c2.txt
pmc c2w "--steps 10 --warmup 2" WRITE_SIZE | tee -a $E/pmc_c2.txt
pmc c2v "--steps 10 --warmup 2" SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU | tee -a $E/pmc_c2.txt
pmc c5f "--workload c5 --steps 3 --warmup 1" FETCH_SIZE | tee $E/pmc_c5.txt
pmc c5w "--workload c5 --steps 3 --warmup 1" WRITE_SIZE | tee -a $E/pmc_c5.txt
pmc c5v "--workload c5 --steps 3 --warmup 1" SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES | tee -a $E/pmc_c5.txt
pmc c5m "--workload c5 --steps 3 --warmup 1" SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM | tee -a $E/pmc_c5.txt
echo "elapsed $(( $(date +%s) - t0 )) s"
echo "== loopback rank 0 of N (C2, automatic exchange interval)"; timeout 400 python tools/shard_loopback.py 2>&1 | grep -v "$F" | tail -4 | tee $E/loopback_c2.txt
echo "== time-sliced child (32 queues + 16 streams), 200 sets"; timeout 600 python tests/c5_timeslice_child.py 200 0 2>&1 | grep -v "$F" | tail -1 | tee $E/timeslice_child.json | cut -c1-300
echo "elapsed $(( $(date +%s) - t0 )) s"
