#!/bin/bash
# Round-3 evidence run: the driver's command (full JSON line incl. CPU baselines and the parity leg), steady state, other configs,
# rocprofv3 kernel stats of the SAME commands, HBM traffic per (workload, workgroups per launch), SQ counters.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
E=gpurun_out/ev3; mkdir -p $E
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d valu_frac %.4f parity %s host %s rowserial %s/%s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks'], d['roofline_valu']['frac'], d.get('parity_ok'), d.get('drawtest_host_ms'), d.get('row_serial_Mray_s'), d.get('row_serial_batched_32_Mray_s')))"; }
echo "== driver's command (full line)"; timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | tee $E/bench_c2_driver_cmd.json | summ
echo "== steady state"; timeout 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | tail -1 | tee $E/bench_c2_steps200.json | summ
for wl in c3 c5 c1; do echo "== $wl"; timeout 600 python bench.py --no-cpu-baseline --no-extras --workload $wl --steps 20 --warmup 10 --parity-frames $([ $wl = c3 ] && echo 0 || echo 64) 2>/dev/null | tail -1 | tee $E/bench_$wl.json | summ; done
echo "== c5 brute force"; timeout 300 python bench.py --no-cpu-baseline --no-extras --workload c5 --steps 6 --warmup 2 --hit-spheres 2 2>/dev/null | tail -1 | tee $E/bench_c5_bruteforce.json | summ
echo "== c2 packed VALU filter (--hit-spheres 3)"; timeout 300 python bench.py --no-cpu-baseline --no-extras --hit-spheres 3 --parity-frames 0 2>/dev/null | tail -1 | tee $E/bench_c2_valu_filter.json | summ
echo "== animate"; timeout 300 python bench.py --no-cpu-baseline --no-extras --animate 2>/dev/null | tail -1 | tee $E/bench_c2_animate.json | summ
echo "== lane-refill kernel"; timeout 300 python bench.py --no-cpu-baseline --no-extras --persistent 1 --parity-frames 0 2>/dev/null | tail -1 | tee $E/bench_c2_persist1.json | summ
echo "== one frame in flight"; timeout 300 python bench.py --no-cpu-baseline --no-extras --overlap 1 --steps 50 --parity-frames 0 2>/dev/null | tail -1 | tee $E/bench_c2_overlap1.json | summ
prof() { # name, args
  cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$E/prof_$1" -o k -- python3 "$R/bench.py" $2 --no-cpu-baseline --no-extras --parity-frames 0 > "$R/$E/prof_$1_bench.json" 2>/dev/null
  cd "$R"; cp $E/prof_$1/k_kernel_stats.csv $E/prof_$1_kernel_stats.csv; head -5 $E/prof_$1_kernel_stats.csv | cut -c1-220; tail -1 $E/prof_$1_bench.json | summ; rm -rf $E/prof_$1
}
echo "== rocprofv3 kernel stats, driver's command"; prof driver_cmd "--gpus 1 --steps 20 --warmup 5"
echo "== rocprofv3 kernel stats, steps 200"; prof steps200 "--steps 200 --warmup 20"
echo "== rocprofv3 kernel stats, c3"; prof c3 "--workload c3 --steps 20 --warmup 10"
echo "== rocprofv3 kernel stats, c5"; prof c5 "--workload c5 --steps 20 --warmup 10"
for wl in c2 c3 c5; do for div in 8 4; do
  extra=""; [ $wl != c2 ] && extra="--steps 4 --warmup 2"
  echo "== traffic $wl griddiv $div"; TPT_GRID_DIV=$div bash tools/traffic.sh "--no-extras --parity-frames 0 --workload $wl $extra" 2>&1 | grep Trace
done; done
echo "== traffic c2 full grid"; bash tools/traffic.sh "--no-extras --parity-frames 0" 2>&1 | grep Trace
echo "== SQ counters at the steady-state grid"; TPT_GRID_DIV=8 bash tools/gpu_pmc.sh "--overlap 1 --no-extras --parity-frames 0" r03 2>&1 | tail -25
echo "== SQ counters c3"; TPT_GRID_DIV=8 bash tools/gpu_pmc.sh "--overlap 1 --no-extras --parity-frames 0 --workload c3 --steps 3 --warmup 1" r03c3 2>&1 | tail -25
echo "== MFMA counters"; cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d "$R/gpurun_out/pmc_mfma" -o p -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --overlap 1 --no-extras --parity-frames 0 > /dev/null 2>&1; cd "$R"; python - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/pmc_mfma/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'Trace' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()): print('%-28s mean %16.1f  n %d' % (k, sum(v)/len(v), len(v)))
PY
