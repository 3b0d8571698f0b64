#!/bin/bash
# Round-2 evidence, final host runtime (paced host, two colour slots per stream, burst-sized grids): the driver's command
# (full JSON line incl. CPU baselines), steady state, the other configs, rocprofv3 kernel stats of the SAME commands, and
# rank 0 of N through the loopback communicator.  (Kernel unchanged since gpu_evidence.sh: its PMC passes still stand.)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
mkdir -p gpurun_out/ev2
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms pipe %.4f grid %d host %s rowserial %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['pipeline_ms_per_step'], d['config']['grid_blocks'], d.get('drawtest_host_ms'), d.get('row_serial_Mray_s')))"; }
echo "== driver's command (full line)"; timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | tee gpurun_out/ev2/bench_c2_driver_cmd.json | summ
echo "== driver's command again"; timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | tee gpurun_out/ev2/bench_c2_driver_cmd_2.json | summ
echo "== steady state"; timeout 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | tail -1 | tee gpurun_out/ev2/bench_c2_steps200.json | summ
echo "== steady state, 600 steps"; timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 600 --warmup 20 2>/dev/null | tail -1 | tee gpurun_out/ev2/bench_c2_steps600.json | summ
echo "== c3"; timeout 300 python bench.py --no-cpu-baseline --no-extras --workload c3 --steps 60 --warmup 10 2>/dev/null | tail -1 | tee gpurun_out/ev2/bench_c3.json | summ
echo "== c5"; timeout 300 python bench.py --no-cpu-baseline --no-extras --workload c5 --steps 40 --warmup 10 2>/dev/null | tail -1 | tee gpurun_out/ev2/bench_c5.json | summ
echo "== c1"; timeout 300 python bench.py --no-cpu-baseline --no-extras --workload c1 --steps 400 --warmup 40 2>/dev/null | tail -1 | tee gpurun_out/ev2/bench_c1.json | summ
echo "== animate"; timeout 300 python bench.py --no-cpu-baseline --no-extras --animate 2>/dev/null | tail -1 | tee gpurun_out/ev2/bench_c2_animate.json | summ
echo "== one frame in flight"; timeout 300 python bench.py --no-cpu-baseline --no-extras --overlap 1 --steps 50 2>/dev/null | tail -1 | tee gpurun_out/ev2/bench_c2_overlap1.json | summ
echo "== lane-refill kernel"; timeout 300 python bench.py --no-cpu-baseline --no-extras --persistent 1 2>/dev/null | tail -1 | tee gpurun_out/ev2/bench_c2_persist1.json | summ
echo "== rocprofv3 kernel stats, driver's command"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/ev2/prof_driver_cmd" -o c2 -- python3 "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > "$R/gpurun_out/ev2/prof_driver_cmd_bench.json" 2>/dev/null
cd "$R"; head -6 gpurun_out/ev2/prof_driver_cmd/c2_kernel_stats.csv; tail -1 gpurun_out/ev2/prof_driver_cmd_bench.json | summ
echo "== rocprofv3 kernel stats, steps 200"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/ev2/prof_steps200" -o c2 -- python3 "$R/bench.py" --steps 200 --warmup 20 --no-cpu-baseline --no-extras > "$R/gpurun_out/ev2/prof_steps200_bench.json" 2>/dev/null
cd "$R"; head -4 gpurun_out/ev2/prof_steps200/c2_kernel_stats.csv; tail -1 gpurun_out/ev2/prof_steps200_bench.json | summ
for wl in c3 c5; do
echo "== rocprofv3 kernel stats, $wl"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/ev2/prof_$wl" -o $wl -- python3 "$R/bench.py" --workload $wl --steps 40 --warmup 10 --no-cpu-baseline --no-extras > "$R/gpurun_out/ev2/prof_${wl}_bench.json" 2>/dev/null
cd "$R"; head -4 gpurun_out/ev2/prof_$wl/${wl}_kernel_stats.csv; tail -1 gpurun_out/ev2/prof_${wl}_bench.json | summ; done
echo "== rank 0 of N (loopback communicator)"; timeout 300 python tools/shard_loopback.py 2>&1 | grep "^N="
find gpurun_out/ev2 -name "*_kernel_trace.csv" -delete; find gpurun_out/ev2 -name "*agent_info*" -delete
