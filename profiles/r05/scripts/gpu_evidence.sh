#!/bin/bash
# Round-5 evidence run on the final tree: full GPU suite (twice), the driver's command with the full JSON line (CPU baselines, parity
# leg), steady state, 30 / 100 frames, C3 / C5 / C1, rocprofv3 kernel stats of the same commands, what the launch writes without its
# pixels (bounce-stack spills alone), the one-GPU loopback table of the sharded path.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
E=gpurun_out/ev5; mkdir -p $E
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); rv=d['roofline_valu']; print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d valu %s %.4f parity %s host %s sync %s rowserial %s/%s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks'], rv['bound'], rv['frac'] or 0, d.get('parity_ok'), d.get('drawtest_host_ms'), d.get('sync_device_caller_ms'), d.get('row_serial_Mray_s'), d.get('row_serial_batched_32_Mray_s')))"; }
t0=$(date +%s)
export TPT_ORACLE_LOG=$PWD/$E/oracle_disagreements.log TPT_MISMATCH_DUMP=$PWD/$E/dump
for i in 1 2; do
  echo "== full GPU suite, run $i"; timeout 600 python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | grep "passed\|failed\|AssertionError\|^E   \|saved\|Error" | cut -c1-300 | head -30
  echo "elapsed $(( $(date +%s) - t0 )) s"
done
echo "== checker disagreements logged:"; cat $E/oracle_disagreements.log 2>/dev/null | cut -c1-300 | head -10; echo "(end)"
unset TPT_MISMATCH_DUMP
echo "== driver's command (full line)"; timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | tee $E/bench_c2_driver_cmd.json | summ
echo "== driver's command again x2 (no extras)"; for i in 1 2; do timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | summ; done
echo "== steady state"; timeout 200 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --steps 200 --warmup 20 2>/dev/null | tail -1 | tee $E/bench_c2_steps200.json | summ
for n in 30 100; do echo "== steps $n"; timeout 200 python bench.py --no-cpu-baseline --no-extras --steps $n --warmup 5 2>/dev/null | tail -1 | tee $E/bench_c2_steps$n.json | summ; done
echo "== c3 (steady)"; timeout 200 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --workload c3 --steps 40 --warmup 20 2>/dev/null | tail -1 | tee $E/bench_c3.json | summ
echo "== c3, one frame, parity leg"; timeout 300 python bench.py --no-cpu-baseline --no-extras --workload c3 --prime 0 --warmup 0 --steps 1 2>/dev/null | tail -1 | tee $E/bench_c3_one_frame_parity.json | summ
echo "== c5"; timeout 200 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --workload c5 --steps 40 --warmup 20 2>/dev/null | tail -1 | tee $E/bench_c5.json | summ
echo "== c5 through the C-ABI exchange at N = 1"; timeout 300 python bench.py --no-cpu-baseline --no-extras --workload c5 --exchange cabi --steps 20 --warmup 10 --parity-frames 0 2>/dev/null | grep '^{"metric"' | tail -1 | tee $E/bench_c5_cabi.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['exchange'], d['rccl_ranks'], d.get('image_fnv'))"
echo "== c1"; timeout 200 python bench.py --no-cpu-baseline --no-extras --workload c1 --steps 200 --warmup 20 --parity-frames 0 2>/dev/null | tail -1 | tee $E/bench_c1.json | summ
echo "== tail helpers off"; TPT_TAIL_HELPERS=0 timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | tee $E/bench_c2_driver_cmd_helpers_off.json | summ
echo "elapsed $(( $(date +%s) - t0 )) s"
prof() { # name, args
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$E/prof_$1" -o k -- python3 "$R/bench.py" $2 --no-cpu-baseline --no-extras --parity-frames 0 > "$R/$E/prof_$1_bench.json" 2>/dev/null
  cd "$R"; cp $E/prof_$1/k_kernel_stats.csv $E/prof_$1_kernel_stats.csv; head -4 $E/prof_$1_kernel_stats.csv | cut -c1-200; tail -1 $E/prof_$1_bench.json | summ; rm -rf $E/prof_$1
}
echo "== rocprofv3 kernel stats, driver's command"; prof driver_cmd "--gpus 1 --steps 20 --warmup 5"
echo "== rocprofv3 kernel stats, steps 200"; prof steps200 "--steps 200 --warmup 20"
echo "== rocprofv3 kernel stats, c5"; prof c5 "--workload c5 --steps 20 --warmup 10"
echo "== WRITE_SIZE of a C2 launch without its pixel stores (measurement build): what is left are the bounce-stack spills"
cd /tmp && TPT_LIB_DIR=$R/tools/_variants/nocolour TPT_GRID_DIV=8 timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$R/gpurun_out/pmc_nocolour" -o p -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --overlap 1 --no-extras --parity-frames 0 > /dev/null 2>&1
cd "$R"; python - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/pmc_nocolour/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'TraceQueue' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()): print('no pixel stores: %-12s mean %12.1f KiB  n %d' % (k, sum(v)/len(v), len(v)))
PY
rm -rf gpurun_out/pmc_nocolour
echo "== loopback: rank 0 of N through the C ABI, frame by frame"; TPT_EMU_BATCH=1 TPT_EMU_N=1,2,4,8 TPT_EMU_FRAMES=320 timeout 200 python tools/shard_loopback.py 2>&1 | grep "^N="
echo "elapsed $(( $(date +%s) - t0 )) s"
