"""Grouped traversal vs brute force around the grouping threshold (256 spheres): Mray/s at 1280x720x4spp."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from toypathtracer_amd import api
from toypathtracer_amd.scenes import STRESS_CAMERA, stress_scene
api.InitializeTest()
w, h, frames, warm = 1280, 720, 60, 20
tile = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
for n, grid in ((128, 12), (256, 16), (512, 24), (1024, 32), (2048, 48)):
    s, m = stress_scene(n, grid)
    api.set_scene(s, m); api.set_camera(**STRESS_CAMERA)
    line = "%5d spheres:" % n
    for hs in (0, 2):
        api.set_kernel_variant(hs, 3, -1)
        for f in range(warm):
            api.UpdateTest(0.0, f, w, h, 2); api.draw_device(0.0, f, w, h, tile.data_ptr(), 2)
        r0 = api.ray_counter_read(); t0 = time.perf_counter()
        for f in range(warm, warm + frames):
            api.UpdateTest(0.0, f, w, h, 2); api.draw_device(0.0, f, w, h, tile.data_ptr(), 2)
        rays = api.ray_counter_read() - r0; dt = time.perf_counter() - t0
        line += "  %s %.2f Gray/s" % ("grouped" if hs == 0 else "brute force", rays / dt / 1e9)
    print(line, flush=True)
api.ShutdownTest()
