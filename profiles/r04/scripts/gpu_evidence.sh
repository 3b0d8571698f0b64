#!/bin/bash
# Round-4 evidence run on the final tree: the driver's command (full JSON line incl. CPU baselines and the parity leg), steady
# state, steps 30 / 100, the other configs, rocprofv3 kernel stats of the SAME commands, HBM traffic per (workload, workgroups per
# launch), SQ / MFMA counters, the one-GPU loopback table of the sharded path.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
E=gpurun_out/ev4; mkdir -p $E
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d valu_frac %.4f parity %s host %s sync %s rowserial %s/%s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks'], d['roofline_valu']['frac'], d.get('parity_ok'), d.get('drawtest_host_ms'), d.get('sync_device_caller_ms'), d.get('row_serial_Mray_s'), d.get('row_serial_batched_32_Mray_s')))"; }
echo "== driver's command (full line)"; timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | tee $E/bench_c2_driver_cmd.json | summ
echo "== steady state"; timeout 200 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --steps 200 --warmup 20 2>/dev/null | tail -1 | tee $E/bench_c2_steps200.json | summ
for n in 30 100; do echo "== steps $n"; timeout 200 python bench.py --no-cpu-baseline --no-extras --steps $n --warmup 5 2>/dev/null | tail -1 | tee $E/bench_c2_steps$n.json | summ; done
echo "== c3 (steady)"; timeout 200 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --workload c3 --steps 40 --warmup 20 2>/dev/null | tail -1 | tee $E/bench_c3.json | summ
echo "== c3, one frame, parity leg"; timeout 300 python bench.py --no-cpu-baseline --no-extras --workload c3 --prime 0 --warmup 0 --steps 1 2>/dev/null | tail -1 | tee $E/bench_c3_one_frame_parity.json | summ
echo "== c3 through the C-ABI exchange at N = 1 (real one-rank RCCL communicator), one frame, parity leg"; timeout 300 python bench.py --no-cpu-baseline --no-extras --workload c3 --exchange cabi --prime 0 --warmup 0 --steps 1 2>/dev/null | grep '^{"metric"' | tail -1 | tee $E/bench_c3_cabi_one_frame_parity.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['exchange'], d['rccl_ranks'], d.get('parity_ok'), d.get('image_fnv'))"
echo "== c5 through the C-ABI exchange at N = 1"; timeout 300 python bench.py --no-cpu-baseline --no-extras --workload c5 --exchange cabi --steps 20 --warmup 10 --parity-frames 0 2>/dev/null | grep '^{"metric"' | tail -1 | tee $E/bench_c5_cabi.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['exchange'], d['rccl_ranks'], d.get('image_fnv'))"
echo "== c5"; timeout 200 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --workload c5 --steps 40 --warmup 20 2>/dev/null | tail -1 | tee $E/bench_c5.json | summ
echo "== c1"; timeout 200 python bench.py --no-cpu-baseline --no-extras --workload c1 --steps 200 --warmup 20 --parity-frames 0 2>/dev/null | tail -1 | tee $E/bench_c1.json | summ
echo "== c2 packed VALU filter (--hit-spheres 3)"; timeout 200 python bench.py --no-cpu-baseline --no-extras --hit-spheres 3 --parity-frames 0 2>/dev/null | tail -1 | tee $E/bench_c2_valu_filter.json | summ
echo "== animate"; timeout 200 python bench.py --no-cpu-baseline --no-extras --animate 2>/dev/null | tail -1 | tee $E/bench_c2_animate.json | summ
echo "== lane-refill kernel"; timeout 200 python bench.py --no-cpu-baseline --no-extras --persistent 1 --parity-frames 0 2>/dev/null | tail -1 | tee $E/bench_c2_persist1.json | summ
echo "== one frame in flight"; timeout 200 python bench.py --no-cpu-baseline --no-extras --overlap 1 --steps 50 --parity-frames 0 2>/dev/null | tail -1 | tee $E/bench_c2_overlap1.json | summ
prof() { # name, args
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$E/prof_$1" -o k -- python3 "$R/bench.py" $2 --no-cpu-baseline --no-extras --parity-frames 0 > "$R/$E/prof_$1_bench.json" 2>/dev/null
  cd "$R"; cp $E/prof_$1/k_kernel_stats.csv $E/prof_$1_kernel_stats.csv; head -4 $E/prof_$1_kernel_stats.csv | cut -c1-200; tail -1 $E/prof_$1_bench.json | summ; rm -rf $E/prof_$1
}
echo "== rocprofv3 kernel stats, driver's command"; prof driver_cmd "--gpus 1 --steps 20 --warmup 5"
echo "== rocprofv3 kernel stats, steps 200"; prof steps200 "--steps 200 --warmup 20"
echo "== rocprofv3 kernel stats, c3"; prof c3 "--workload c3 --steps 20 --warmup 10"
echo "== rocprofv3 kernel stats, c5"; prof c5 "--workload c5 --steps 20 --warmup 10"
for wl in c2 c3 c5; do
  extra=""; [ $wl != c2 ] && extra="--steps 4 --warmup 2"
  echo "== traffic $wl griddiv 8"; TPT_GRID_DIV=8 bash tools/traffic.sh "--no-extras --parity-frames 0 --workload $wl $extra" 2>&1 | grep Trace
done
echo "== SQ counters at the steady-state grid, c2"; TPT_GRID_DIV=8 bash tools/gpu_pmc.sh "--overlap 1 --no-extras --parity-frames 0" r04 2>&1 | tail -25
echo "== SQ counters c5"; cd /tmp && TPT_GRID_DIV=8 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d "$R/gpurun_out/pmc_r04c5" -o p -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --overlap 1 --no-extras --parity-frames 0 --workload c5 > /dev/null 2>&1; cd "$R"; python - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/pmc_r04c5/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'Trace' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()): print('%-28s mean %16.1f  n %d' % (k, sum(v)/len(v), len(v)))
PY
echo "== MFMA counters"; cd /tmp && TPT_GRID_DIV=8 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d "$R/gpurun_out/pmc_mfma4" -o p -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --overlap 1 --no-extras --parity-frames 0 > /dev/null 2>&1; cd "$R"; python - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/pmc_mfma4/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'Trace' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()): print('%-28s mean %16.1f  n %d' % (k, sum(v)/len(v), len(v)))
PY
echo "== loopback: rank 0 of N through the C ABI, frame by frame (stream batching on = default)"; TPT_EMU_BATCH=1 TPT_EMU_N=1,2,4,8 TPT_EMU_FRAMES=320 timeout 200 python tools/shard_loopback.py 2>&1 | grep "^N="
echo "== loopback, stream batching off"; TPT_STREAM_BATCH=0 TPT_EMU_BATCH=1 TPT_EMU_N=1,2,4,8 TPT_EMU_FRAMES=320 timeout 200 python tools/shard_loopback.py 2>&1 | grep "^N="
echo "== loopback, 4 / 8 frames per launch and exchange"; for b in 4 8; do TPT_EMU_BATCH=$b TPT_EMU_N=1,8 TPT_EMU_FRAMES=320 timeout 200 python tools/shard_loopback.py 2>&1 | grep "^N="; done
