"""Host-side enqueue cost per frame (tiny image: the GPU work is negligible, the loop is bound by the host)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from toypathtracer_amd import api
api.InitializeTest()
lib = api.load_library()
w, h, n = 64, 64, 3000
tile = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
CASES = [(3, int(o)) for o in os.environ["TPT_OV"].split(",")] if os.environ.get("TPT_OV") else ((3, 1), (3, 2), (3, 16), (1, 1), (1, 16))
for persist, ov in CASES:
    api.set_kernel_variant(0, persist, -1)
    api.set_frame_overlap(ov)
    for f in range(200):
        api.UpdateTest(0.0, f, w, h, 2); api.draw_device(0.0, f, w, h, tile.data_ptr(), 2)
    api.synchronize()
    api.kernel_timing_begin(n)
    t0 = time.perf_counter()
    for f in range(n):
        api.UpdateTest(0.0, f, w, h, 2); api.draw_device(0.0, f, w, h, tile.data_ptr(), 2)
    t1 = time.perf_counter()
    api.synchronize()
    t2 = time.perf_counter()
    ms, cnt = api.kernel_timing_end()
    print("kernel %d overlap %2d: enqueue %.1f us/frame, incl. drain %.1f us/frame, trace launch %.1f us" % (persist, ov, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6, ms / cnt * 1e3))
t0 = time.perf_counter()
for f in range(20000):
    lib.tptGetLastError()
print("ctypes call: %.2f us" % ((time.perf_counter() - t0) / 20000 * 1e6))
t0 = time.perf_counter()
for f in range(3000):
    api.UpdateTest(0.0, f, w, h, 2)
print("UpdateTest alone: %.2f us" % ((time.perf_counter() - t0) / 3000 * 1e6))
api.ShutdownTest()
