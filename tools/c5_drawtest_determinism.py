"""DrawTest(host buffer) of frame 0 of the 4096-sphere scene over and over, the way tests/test_gpu_parity.py's gpu_frames calls it
(look-ahead frames traced behind each call and dropped by the next): ray count and image must not change."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
import numpy as np  # noqa: E402

from oracle_lib import fnv1a  # noqa: E402
from toypathtracer_amd import api as tpt  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
w, h, spp = 1920, 1080, 8
tpt.InitializeTest()
from toypathtracer_amd.scenes import STRESS_CAMERA, stress_scene  # noqa: E402
s, m = stress_scene(4096, 64)
tpt.set_scene(s, m)
tpt.set_camera(**STRESS_CAMERA)
tpt.set_samples_per_pixel(spp)
seen = {}
for r in range(reps):
    bb = np.zeros((h, w, 4), np.float32)
    tpt.UpdateTest(0.0, 0, w, h, 2)
    rays = tpt.DrawTest(0.0, 0, w, h, bb, 2)
    key = (rays, "%08x" % fnv1a(bb))
    seen[key] = seen.get(key, 0) + 1
    if seen[key] == 1:
        print("rep %d: rays %d image %s" % (r, rays, key[1]), flush=True)
print("results:", seen)
tpt.ShutdownTest()
