"""What do extra streams in the host process cost?  N torch streams (each used once, then idle), then the steady-state C2 stream of
frames.  Round 4 measured 60 -> 25 Gray/s with four extra streams at GPU_MAX_HW_QUEUES=32 (the process then held more queues than the
device runs side by side); with the round-5 default of 20 the extra streams share queues instead.
    GPU_MAX_HW_QUEUES=20 python tools/extra_streams_rate.py 8"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
import torch  # noqa: E402

from toypathtracer_amd import api  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
api.InitializeTest()
extra = []
for _ in range(n):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        torch.zeros(16, device="cuda").add_(1.0)
    extra.append(st)
torch.cuda.synchronize()
w, h = 1280, 720
tile = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")


def burst(f0, k):
    r0 = api.ray_counter_read()
    t0 = time.perf_counter()
    for f in range(f0, f0 + k):
        api.UpdateTest(0.0, f, w, h, 2)
        api.draw_device(0.0, f, w, h, tile.data_ptr(), 2)
    rays = api.ray_counter_read() - r0
    return rays / (time.perf_counter() - t0) / 1e6


burst(0, 30)
rates = [burst(30 + 100 * i, 100) for i in range(3)]
print("GPU_MAX_HW_QUEUES=%s, %d extra streams: %s Mray/s  %s" % (os.environ["GPU_MAX_HW_QUEUES"], n, " ".join("%.0f" % r for r in rates), api.pipeline_info()))
api.ShutdownTest()
