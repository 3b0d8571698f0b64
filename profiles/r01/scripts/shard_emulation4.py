"""Per-rank render time for N-way row sharding of C2 with the path-queue kernel (render only, no gather):
aggregate = rays of rank 0's tile x N / time.  Env TPT_GRID_DIV fixes the workgroups-per-launch divisor (default adaptive)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
from toypathtracer_amd import api
api.InitializeTest()
w, h, frames, warm = 1280, 720, 400, 60
for n in (1, 2, 4, 8):
    line = "N=%d:" % n
    for ov in (8, 16):
        api.set_frame_overlap(ov)
        api.set_row_shard(8, n, 0)
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
        info = api.launch_info() if hasattr(api, "launch_info") else {}
        line += "  ov%d %.3f ms (%.1f G, grid %s)" % (ov, dt / frames * 1e3, rays / dt / 1e9 * n, info.get("grid_blocks", "?"))
    print(line, flush=True)
api.set_row_shard(0, 1, 0)
api.ShutdownTest()
