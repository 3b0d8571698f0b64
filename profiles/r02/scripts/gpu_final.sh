#!/bin/bash
# Round-2 final run on the committed tree: smoke(), the whole GPU suite, the bench lines the documents quote, rocprofv3
# kernel stats of the driver's command and of the steady state (same commands), rank 0 of N with 1 / 4 frames per launch.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
mkdir -p gpurun_out/ev3
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -5
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms pipe %.4f grid %d host %s rowserial %s batched4 %s batched8 %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['pipeline_ms_per_step'], d['config']['grid_blocks'], d.get('drawtest_host_ms'), d.get('row_serial_Mray_s'), d.get('batched_4_Mray_s'), d.get('batched_8_Mray_s')))"; }
echo "== driver's command (full line)"; timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | tee gpurun_out/ev3/bench_c2_driver_cmd.json | summ
echo "== steady state"; timeout 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | tail -1 | tee gpurun_out/ev3/bench_c2_steps200.json | summ
echo "== c3, 60 frames"; timeout 300 python bench.py --no-cpu-baseline --workload c3 --steps 60 --warmup 10 2>/dev/null | tail -1 | tee gpurun_out/ev3/bench_c3.json | summ
echo "== c1"; timeout 300 python bench.py --no-cpu-baseline --workload c1 --steps 400 --warmup 40 2>/dev/null | tail -1 | tee gpurun_out/ev3/bench_c1.json | summ
echo "== c2, 4 frames per launch"; timeout 300 python bench.py --no-cpu-baseline --no-extras --batch 4 --steps 200 --warmup 20 2>/dev/null | tail -1 | tee gpurun_out/ev3/bench_c2_batch4.json | summ
echo "== rocprofv3 kernel stats, driver's command"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/ev3/prof_driver_cmd" -o c2 -- python3 "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > "$R/gpurun_out/ev3/prof_driver_cmd_bench.json" 2>/dev/null
cd "$R"; head -5 gpurun_out/ev3/prof_driver_cmd/c2_kernel_stats.csv; tail -1 gpurun_out/ev3/prof_driver_cmd_bench.json | summ
echo "== rocprofv3 kernel stats, steps 200"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/ev3/prof_steps200" -o c2 -- python3 "$R/bench.py" --steps 200 --warmup 20 --no-cpu-baseline --no-extras > "$R/gpurun_out/ev3/prof_steps200_bench.json" 2>/dev/null
cd "$R"; head -4 gpurun_out/ev3/prof_steps200/c2_kernel_stats.csv; tail -1 gpurun_out/ev3/prof_steps200_bench.json | summ
echo "== rocprofv3 kernel stats, 4 frames per launch"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/ev3/prof_batch4" -o c2 -- python3 "$R/bench.py" --batch 4 --steps 200 --warmup 20 --no-cpu-baseline --no-extras > "$R/gpurun_out/ev3/prof_batch4_bench.json" 2>/dev/null
cd "$R"; head -4 gpurun_out/ev3/prof_batch4/c2_kernel_stats.csv; tail -1 gpurun_out/ev3/prof_batch4_bench.json | summ
for b in 1 4; do echo "== rank 0 of N, $b frame(s) per launch"; TPT_EMU_BATCH=$b TPT_EMU_FRAMES=320 timeout 200 python tools/shard_loopback.py 2>&1 | grep "^N="; done
find gpurun_out/ev3 -name "*_kernel_trace.csv" -delete; find gpurun_out/ev3 -name "*agent_info*" -delete
