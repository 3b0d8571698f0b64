#!/bin/bash
# r02 run 26: host-pointer DrawTest: page-locked buffer, trusted-buffer mode, look-ahead
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
echo "== host path tests"
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py -m gpu -q -x -k "drawtest or cxx_host or golden or config2 or sharded_equals" 2>&1 | tail -5
echo "== host path timing"
timeout 200 python - <<'PY'
import time, numpy as np
from toypathtracer_amd import api
api.InitializeTest()
w, h = 1280, 720
for (la, trust) in [(0, 0), (1, 0), (2, 0), (3, 0), (0, 1), (2, 1), (3, 1)]:
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
echo "== full gpu suite"
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8
