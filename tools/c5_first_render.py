"""The pattern of tests/test_gpu_parity.py::test_config5_stress_scene_full_parity that came back with one wrong pixel in the FIRST render
after the scene change (round 5, 2 of 4 suite runs): default scene -> a small frame -> 4096-sphere scene -> DrawTest of the full frame,
over and over; every full frame must hash alike."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
import numpy as np  # noqa: E402

from oracle_lib import fnv1a  # noqa: E402
from toypathtracer_amd import api as tpt  # noqa: E402
from toypathtracer_amd.scenes import STRESS_CAMERA, stress_scene  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
w, h, spp = 1920, 1080, 8
tpt.InitializeTest()
s, m = stress_scene(4096, 64)
seen = {}
for r in range(reps):
    tpt.set_scene(None)
    tpt.set_camera(None)
    tpt.set_samples_per_pixel(4)
    small = np.zeros((90, 160, 4), np.float32)
    for f in range(3):
        tpt.UpdateTest(0.0, f, 160, 90, 2)
        tpt.DrawTest(0.0, f, 160, 90, small, 2)
    tpt.set_scene(s, m)
    tpt.set_camera(**STRESS_CAMERA)
    tpt.set_samples_per_pixel(spp)
    out = []
    for k in range(2):
        bb = np.zeros((h, w, 4), np.float32)
        tpt.UpdateTest(0.0, 0, w, h, 2)
        rays = tpt.DrawTest(0.0, 0, w, h, bb, 2)
        out.append((rays, "%08x" % fnv1a(bb)))
    for key in out:
        seen[key] = seen.get(key, 0) + 1
    if len(set(out)) > 1 or len(seen) > 1:
        print("rep %d: %s" % (r, out), flush=True)
print("results:", seen)
tpt.ShutdownTest()
