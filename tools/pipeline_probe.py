"""Host-side view of the frame pipeline: time spent in each tptDrawDevice call, grid and pipeline depth per frame."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
import torch
from toypathtracer_amd import api
api.InitializeTest()
print("pipeline:", api.pipeline_info())
w, h = 1280, 720
tile = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
s = torch.cuda.Stream()
api.set_stream(s.cuda_stream)
n = int(os.environ.get("N", "40"))
for rep in range(2):
    ts, grids, depth = [], [], []
    t00 = time.perf_counter()
    for f in range(n):
        t0 = time.perf_counter()
        api.UpdateTest(0.0, f, w, h, 2)
        api.draw_device(0.0, f, w, h, tile.data_ptr(), 2)
        ts.append((time.perf_counter() - t0) * 1e6)
        grids.append(api.launch_info()["grid_blocks"])
        depth.append(api.pipeline_info()["stream_depth"])
    t1 = time.perf_counter()
    api.synchronize()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("rep %d: enqueue %.2f ms, total %.2f ms (%.3f ms/frame)" % (rep, (t1 - t00) * 1e3, (t2 - t00) * 1e3, (t2 - t00) / n * 1e3))
    print("  host us per call:", " ".join("%.0f" % x for x in ts))
    print("  grid:", grids)
    print("  depth:", depth)
api.set_stream(None)
api.ShutdownTest()
