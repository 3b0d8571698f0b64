"""Per-rank render time when the frame is sharded N ways (row stripes of 8), measured on ONE GPU by rendering
rank r's tile only -- predicts the render part of the N-GPU strong-scaling curve (the gather is not included)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
from toypathtracer_amd import api
api.InitializeTest()
w, h, frames, warm = 1280, 720, 200, 30
for n in (1, 2, 4, 8):
    for r in sorted(set([0, n - 1])):
        api.set_row_shard(8, n, r)
        rows = api.local_row_count(h)
        tile = torch.zeros((rows, w, 4), dtype=torch.float32, device="cuda")
        for f in range(warm):
            api.UpdateTest(0.0, f, w, h, 2); api.draw_device(0.0, f, w, h, tile.data_ptr(), 2)
        r0 = api.ray_counter_read()
        t0 = time.perf_counter()
        for f in range(warm, warm + frames):
            api.UpdateTest(0.0, f, w, h, 2); api.draw_device(0.0, f, w, h, tile.data_ptr(), 2)
        rays = api.ray_counter_read() - r0
        dt = time.perf_counter() - t0
        print("N=%d rank %d: %d rows, %.4f ms/frame, %.1f Mray/s on this rank -> %.1f Mray/s if all %d ranks ran like it (render only)" % (
            n, r, rows, dt / frames * 1e3, rays / dt / 1e6, rays / dt / 1e6 * n, n))
api.set_row_shard(0, 1, 0)
api.ShutdownTest()
