#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
echo "== lifecycle alone"; timeout 300 python -m pytest tests/test_gpu_zz_lifecycle.py -m gpu -q -x 2>&1 | tail -12 | cut -c1-200
echo "== small scenes + lifecycle"; timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_zz_lifecycle.py -m gpu -q -x -k "small_scenes or shutdown" 2>&1 | tail -12 | cut -c1-200
echo "== lifecycle, lookahead 0"; TPT_HOST_LOOKAHEAD=0 timeout 300 python -m pytest tests/test_gpu_zz_lifecycle.py -m gpu -q -x 2>&1 | tail -5 | cut -c1-200
echo "== host path timing"
timeout 200 python - <<'PY'
import time, numpy as np
from toypathtracer_amd import api
api.InitializeTest()
w, h = 1280, 720
for (la, trust) in [(0, 0), (2, 0), (3, 0), (2, 1)]:
    api.set_host_lookahead(la); api.set_host_buffer_mode(trust)
    bb = np.zeros((h, w, 4), np.float32)
    for f in range(6):
        api.UpdateTest(0.0, f, w, h, 2); api.DrawTest(0.0, f, w, h, bb, 2)
    t0 = time.perf_counter(); rays = 0
    for f in range(6, 56):
        api.UpdateTest(0.0, f, w, h, 2); rays += api.DrawTest(0.0, f, w, h, bb, 2)
    dt = time.perf_counter() - t0
    print("lookahead %d trust %d: DrawTest(host backbuffer) %.3f ms/frame, %.1f Mray/s" % (la, trust, dt / 50 * 1e3, rays / dt / 1e6), flush=True)
api.ShutdownTest()
PY
