#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
echo "== pytest gpu"; timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== bench default"; timeout 300 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_c2_default.json | cut -c1-300
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms occ %d grid %d lds %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['blocks_per_cu'], d['config']['grid_blocks'], d['config']['lds_bytes_per_block']))"; }
for args in "--overlap 2" "--fold 1" "--workload c3 --steps 30 --warmup 12" "--workload c5 --steps 10 --warmup 9" "--workload c1"; do echo "-- $args"; timeout 120 python bench.py --no-cpu-baseline $args 2>&1 | tail -1 | summ; done
echo "== torchrun 1 rank"; timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --no-cpu-baseline 2>&1 | tail -1 | summ
echo "== rocprof kernel trace, default command"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_default" -o c2 -- python "$R/bench.py" --no-cpu-baseline > "$R/gpurun_out/prof_default_bench.json" 2>/dev/null
cd "$R"; head -5 gpurun_out/prof_default/c2_kernel_stats.csv; tail -1 gpurun_out/prof_default_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench under rocprof: value', d['value'], 'launch_ms_avg', d['trace_launch_ms_avg'], 'pipeline', d['pipeline_ms_per_step'])"
