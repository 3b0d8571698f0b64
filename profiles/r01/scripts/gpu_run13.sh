#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
cat > /tmp/qtest.py <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, torch
from toypathtracer_amd import api
from oracle_lib import Oracle, SEED_PER_PIXEL
o = Oracle.get()
api.InitializeTest()
api.set_kernel_variant(0, 3, -1)
for (w, h, spp, frames) in [(64, 64, 1, 1), (320, 184, 4, 3), (203, 117, 4, 2), (1280, 720, 4, 2)]:
    api.set_samples_per_pixel(spp)
    bb = np.zeros((h, w, 4), np.float32); per = []
    t0 = time.time()
    for f in range(frames):
        api.UpdateTest(0.0, f, w, h, 2); per.append(api.DrawTest(0.0, f, w, h, bb, 2))
    dt = time.time() - t0
    s, m = o.default_scene(); cam = o.default_camera(w, h); bo = np.zeros((h, w, 4), np.float32); pero = []
    for f in range(frames):
        r, _ = o.render(s, m, cam, w, h, spp, f, seed_mode=SEED_PER_PIXEL, backbuffer=bo); pero.append(r)
    print(w, h, spp, frames, 'rays', per, pero, 'equal', bb.tobytes() == bo.tobytes(), 'maxdiff', float(np.abs(bb - bo).max()), '%.3fs' % dt, flush=True)
api.ShutdownTest()
PY
timeout 60 python /tmp/qtest.py 2>&1 | grep -v amdgpu.ids
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms occ %d grid %d lds %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['blocks_per_cu'], d['config']['grid_blocks'], d['config']['lds_bytes_per_block']))"; }
for args in "--overlap 1" "--overlap 2" "--workload c3 --steps 10 --warmup 2 --overlap 1" "--workload c3 --steps 10 --warmup 2"; do echo "-- queue $args"; timeout 60 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --persistent 3 $args 2>&1 | tail -1 | summ; done
