#!/bin/bash
# r03 closing run on the final tree: full GPU suite, smoke, the driver's command (full line), steady state, C3, rocprofv3 kernel stats
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
E=gpurun_out/ev5; mkdir -p $E
echo "== full GPU suite"; timeout 1500 python -m pytest tests -m gpu -q --tb=line 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -5
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d valu_frac %.4f hbm_frac %.5f traffic %s parity %s host %s sync %s rowserial %s/%s batched %s/%s cpu %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks'], d['roofline_valu']['frac'], d['roofline']['frac'], d['roofline']['traffic'], d.get('parity_ok'), d.get('drawtest_host_ms'), d.get('sync_device_caller_ms'), d.get('row_serial_Mray_s'), d.get('row_serial_batched_32_Mray_s'), d.get('batched_4_Mray_s'), d.get('batched_8_Mray_s'), d.get('cpu_baseline',{}).get('value')))"; }
echo "== driver's command (full line)"; timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | tee $E/bench_c2_driver_cmd.json | summ
echo "== steady state"; timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 20 --parity-frames 0 2>/dev/null | tail -1 | tee $E/bench_c2_steps200.json | summ
echo "== c3"; timeout 300 python bench.py --no-cpu-baseline --no-extras --workload c3 --steps 20 --warmup 10 --parity-frames 0 2>/dev/null | tail -1 | tee $E/bench_c3.json | summ
prof() { cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$E/prof_$1" -o k -- python3 "$R/bench.py" $2 --no-cpu-baseline --no-extras --parity-frames 0 > "$R/$E/prof_$1_bench.json" 2>/dev/null
  cd "$R"; cp $E/prof_$1/k_kernel_stats.csv $E/prof_$1_kernel_stats.csv; head -3 $E/prof_$1_kernel_stats.csv | cut -c1-200; tail -1 $E/prof_$1_bench.json | summ; rm -rf $E/prof_$1; }
echo "== rocprofv3 kernel stats, driver's command"; prof driver_cmd "--gpus 1 --steps 20 --warmup 5"
echo "== rocprofv3 kernel stats, steps 200"; prof steps200 "--steps 200 --warmup 20"
