"""The same frame of a workload rendered over and over: ray count and image hash must not change (round 5: the 4096-sphere
scene's second render came back with 34 more rays once).   python tools/c5_determinism.py [workload] [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
import torch  # noqa: E402

from oracle_lib import fnv1a  # noqa: E402
from toypathtracer_amd import api as tpt  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c5"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
w, h, spp = {"c5": (1920, 1080, 8), "c2": (1280, 720, 4)}[wl]
tpt.InitializeTest()
if wl == "c5":
    from toypathtracer_amd.scenes import STRESS_CAMERA, stress_scene
    s, m = stress_scene(4096, 64)
    tpt.set_scene(s, m)
    tpt.set_camera(**STRESS_CAMERA)
tpt.set_samples_per_pixel(spp)
seen = {}
for r in range(reps):
    frames = 1 if r % 2 == 0 else 3  # (alone, and three frames in flight with a blocking wait: the tail helpers come in)
    tiles = [torch.zeros((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(frames)]
    r0 = tpt.ray_counter_read()
    for f in range(frames):
        tpt.UpdateTest(0.0, f, w, h, 2)
        tpt.draw_device(0.0, f, w, h, tiles[f].data_ptr(), 2)
    tpt.synchronize()
    rays = tpt.ray_counter_read() - r0
    key = " ".join("%08x" % fnv1a(t.cpu().numpy()) for t in tiles)
    print("rep %d (%d frames in flight): rays %d image %s" % (r, frames, rays, key), flush=True)
tpt.ShutdownTest()
