#!/bin/bash
# Round-1 evidence run for the path-queue default: tests, bench (+CPU baseline), other configs, rocprofv3 kernel stats of
# the default command, HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes), SQ counters.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3
echo "== bench default"; timeout 300 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_c2_default.json | cut -c1-400
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms occ %d grid %d lds %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['blocks_per_cu'], d['config']['grid_blocks'], d['config']['lds_bytes_per_block']))"; }
for args in "--overlap 1" "--overlap 2" "--persistent 1" "--persistent 1 --fold 1" "--animate" "--workload c1" "--workload c3 --steps 30 --warmup 20" "--workload c5 --steps 30 --warmup 20" "--workload c5 --steps 20 --warmup 20 --hit-spheres 2"; do echo "-- $args"; timeout 200 python bench.py --no-cpu-baseline $args 2>&1 | tail -1 | summ; done
echo "== rocprof kernel trace, default command"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_default" -o c2 -- python "$R/bench.py" --no-cpu-baseline > "$R/gpurun_out/prof_default_bench.json" 2>/dev/null
cd "$R"; head -5 gpurun_out/prof_default/c2_kernel_stats.csv; tail -1 gpurun_out/prof_default_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench under rocprof: value', d['value'], 'launch_ms_avg', d['trace_launch_ms_avg'], 'pipeline', d['pipeline_ms_per_step'])"
echo "== rocprof kernel trace, 2 frames in flight (per-launch duration undisturbed by queueing)"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_ov2" -o c2 -- python "$R/bench.py" --no-cpu-baseline --overlap 2 > "$R/gpurun_out/prof_ov2_bench.json" 2>/dev/null
cd "$R"; head -3 gpurun_out/prof_ov2/c2_kernel_stats.csv; tail -1 gpurun_out/prof_ov2_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench under rocprof: value', d['value'], 'launch_ms_avg', d['trace_launch_ms_avg'], 'pipeline', d['pipeline_ms_per_step'])"
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c"
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$R/gpurun_out/pmc_$c" -o p -- python "$R/bench.py" --steps 20 --warmup 10 --no-cpu-baseline --overlap 1 > /dev/null 2>&1
  cd "$R"; python - <<PY
import csv, glob, collections
for f in glob.glob('gpurun_out/pmc_$c/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k, v in acc.items(): print(k, 'mean', sum(v)/len(v), 'n', len(v))
PY
done
echo "== SQ counters (single launch in flight, full grid)"
bash tools/gpu_pmc.sh "--overlap 1" r31 2>&1 | tail -25
echo "== SQ counters at the steady-state grid (64 workgroups)"
TPT_GRID_DIV=8 bash tools/gpu_pmc.sh "--overlap 1" r31_div8 2>&1 | tail -25
echo "== smoke"; timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== soak"; timeout 300 python tools/soak.py 2>&1 | tail -3
echo "== sharded exchange emulation"; for n in 1 2 4 8; do TPT_EMU_OV=$([ $n -le 2 ] && echo 16 || echo 8) TPT_EMU_N=$n timeout 100 python tools/shard_exchange_emu.py 2>&1 | grep N=; done
echo "== host-pointer DrawTest"; timeout 100 python - <<'PY'
import time, numpy as np
from toypathtracer_amd import api
api.InitializeTest()
w, h = 1280, 720
bb = np.zeros((h, w, 4), np.float32)
for f in range(5):
    api.UpdateTest(0.0, f, w, h, 2); api.DrawTest(0.0, f, w, h, bb, 2)
t0 = time.perf_counter(); rays = 0
for f in range(5, 45):
    api.UpdateTest(0.0, f, w, h, 2); rays += api.DrawTest(0.0, f, w, h, bb, 2)
dt = time.perf_counter() - t0
print("DrawTest(host backbuffer) %.3f ms/frame, %.1f Mray/s" % (dt / 40 * 1e3, rays / dt / 1e6))
api.ShutdownTest()
PY
