#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
echo "== bench default"; timeout 300 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_c2_default.json
echo "== rocprof kernel trace, default command"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_default" -o c2 -- python "$R/bench.py" --no-cpu-baseline > "$R/gpurun_out/prof_default_bench.json" 2>/dev/null
cd "$R"; head -4 gpurun_out/prof_default/c2_kernel_stats.csv; tail -1 gpurun_out/prof_default_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench under rocprof: value', d['value'], 'launch_ms_avg', d['trace_launch_ms_avg'], 'pipeline', d['pipeline_ms_per_step'])"
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c"
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$R/gpurun_out/pmc_$c" -o p -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --overlap 1 > /dev/null 2>&1
  cd "$R"; python - <<PY
import csv, glob, collections
for f in glob.glob('gpurun_out/pmc_$c/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k, v in acc.items(): print(k, 'mean', sum(v)/len(v), 'n', len(v))
PY
done
echo "== host-pointer DrawTest (PCIe inclusive)"
timeout 120 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import time, numpy as np, torch
from toypathtracer_amd import api
api.InitializeTest()
w, h = 1280, 720
bb = np.zeros((h, w, 4), np.float32)
for f in range(3): api.UpdateTest(0.0, f, w, h, 2); api.DrawTest(0.0, f, w, h, bb, 2)
t0 = time.perf_counter(); rays = 0
for f in range(3, 33): api.UpdateTest(0.0, f, w, h, 2); rays += api.DrawTest(0.0, f, w, h, bb, 2)
dt = time.perf_counter() - t0
print('DrawTest host pointer (pageable numpy): %.3f ms/frame, %.1f Mray/s' % (dt / 30 * 1e3, rays / dt / 1e6))
pin = torch.zeros((h, w, 4), dtype=torch.float32).pin_memory(); pb = pin.numpy()
for f in range(3): api.UpdateTest(0.0, f, w, h, 2); api.DrawTest(0.0, f, w, h, pb, 2)
t0 = time.perf_counter(); rays = 0
for f in range(3, 33): api.UpdateTest(0.0, f, w, h, 2); rays += api.DrawTest(0.0, f, w, h, pb, 2)
dt = time.perf_counter() - t0
print('DrawTest host pointer (pinned): %.3f ms/frame, %.1f Mray/s' % (dt / 30 * 1e3, rays / dt / 1e6))
api.ShutdownTest()
PY
echo "== torchrun single rank"; timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
echo "== c3 / c5 default"; for wl in "c3 --steps 10 --warmup 2" "c5 --steps 5 --warmup 1"; do timeout 120 python bench.py --workload $wl --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:12], d['value'], d['ms_per_step'], d['roofline_valu']['frac_per_pipeline_slot'])"; done
