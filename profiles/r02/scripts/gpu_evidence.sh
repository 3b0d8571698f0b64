#!/bin/bash
# Round-2 evidence run: the driver's command (full JSON line incl. CPU baselines), steady state, other configs, rocprofv3
# kernel stats of the SAME command, HBM traffic and SQ counters at the launch geometry of that command.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
mkdir -p gpurun_out/ev
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms pipe %.4f grid %d host %s rowserial %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['pipeline_ms_per_step'], d['config']['grid_blocks'], d.get('drawtest_host_ms'), d.get('row_serial_Mray_s')))"; }
echo "== driver's command (full line)"; timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | tee gpurun_out/ev/bench_c2_driver_cmd.json | summ
echo "== steady state"; timeout 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | tail -1 | tee gpurun_out/ev/bench_c2_steps200.json | summ
for wl in c3 c5 c1; do echo "== $wl"; timeout 300 python bench.py --no-cpu-baseline --no-extras --workload $wl --steps 20 --warmup 10 2>/dev/null | tail -1 | tee gpurun_out/ev/bench_$wl.json | summ; done
echo "== c5 brute force"; timeout 300 python bench.py --no-cpu-baseline --no-extras --workload c5 --steps 6 --warmup 2 --hit-spheres 2 2>/dev/null | tail -1 | tee gpurun_out/ev/bench_c5_bruteforce.json | summ
echo "== animate"; timeout 300 python bench.py --no-cpu-baseline --no-extras --animate 2>/dev/null | tail -1 | tee gpurun_out/ev/bench_c2_animate.json | summ
echo "== lane-refill kernel"; timeout 300 python bench.py --no-cpu-baseline --no-extras --persistent 1 2>/dev/null | tail -1 | tee gpurun_out/ev/bench_c2_persist1.json | summ
echo "== rocprofv3 kernel stats, driver's command"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/ev/prof_driver_cmd" -o c2 -- python3 "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > "$R/gpurun_out/ev/prof_driver_cmd_bench.json" 2>/dev/null
cd "$R"; head -6 gpurun_out/ev/prof_driver_cmd/c2_kernel_stats.csv; tail -1 gpurun_out/ev/prof_driver_cmd_bench.json | summ
echo "== rocprofv3 kernel stats, steps 200"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/ev/prof_steps200" -o c2 -- python3 "$R/bench.py" --steps 200 --warmup 20 --no-cpu-baseline --no-extras > "$R/gpurun_out/ev/prof_steps200_bench.json" 2>/dev/null
cd "$R"; head -4 gpurun_out/ev/prof_steps200/c2_kernel_stats.csv; tail -1 gpurun_out/ev/prof_steps200_bench.json | summ
for wl in c3 c5; do
echo "== rocprofv3 kernel stats, $wl"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/ev/prof_$wl" -o $wl -- python3 "$R/bench.py" --workload $wl --steps 20 --warmup 10 --no-cpu-baseline --no-extras > "$R/gpurun_out/ev/prof_${wl}_bench.json" 2>/dev/null
cd "$R"; head -4 gpurun_out/ev/prof_$wl/${wl}_kernel_stats.csv; done
echo "== HBM traffic, steady-state grid (64 workgroups per launch)"; TPT_GRID_DIV=8 bash tools/traffic.sh "--no-extras" 2>&1 | grep Trace
echo "== HBM traffic, full grid"; bash tools/traffic.sh "--no-extras" 2>&1 | grep Trace
echo "== HBM traffic c3, steady-state grid"; TPT_GRID_DIV=8 bash tools/traffic.sh "--no-extras --workload c3 --steps 4 --warmup 2" 2>&1 | grep Trace
echo "== SQ counters at the steady-state grid"; TPT_GRID_DIV=8 bash tools/gpu_pmc.sh "--overlap 1 --no-extras" r02 2>&1 | tail -25
echo "== SQ counters c3"; TPT_GRID_DIV=8 bash tools/gpu_pmc.sh "--overlap 1 --no-extras --workload c3 --steps 3 --warmup 1" r02c3 2>&1 | tail -25
