"""Grouped scenes rendered frame after frame: per scene and frame the ray count and the image hash, one line each.  Run once per build
(TPT_LIB_DIR=...) and diff the outputs: the half-line bounds test against the line-only build (-DTPT_DEAL_HALF_LINE=0), the three-stage
dealing at 64-entry areas, the flat filter (--hit-spheres 3) -- independent implementations of the same traversal that must agree bit
for bit.  Scenes: the 4096-sphere field (BASELINE configs[4]), a 20 000-sphere field, a dense 1000-sphere one, and two clouds of spheres
spread through a VOLUME (radii over a factor of 4, a quarter of them metal / glass, eight lights) seen from inside and from outside -- bounds
in every direction around the rays, nothing flat about them.   python tools/grouped_soak.py [frames] [hit_spheres]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle_lib import fnv1a  # noqa: E402
from toypathtracer_amd import api as tpt  # noqa: E402
from toypathtracer_amd.scenes import CLOUD_CAMERA_INSIDE, CLOUD_CAMERA_OUTSIDE, STRESS_CAMERA, cloud_scene, stress_scene  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
hs = int(sys.argv[2]) if len(sys.argv) > 2 else 0
only = sys.argv[3] if len(sys.argv) > 3 else ""


SCENES = [
    ("field 4096", stress_scene(4096, 64), STRESS_CAMERA, 640, 360, 4),
    ("field 20000", stress_scene(20000, 160), STRESS_CAMERA, 480, 270, 2),
    ("field 1000 dense", stress_scene(1000, 20), STRESS_CAMERA, 480, 270, 4),
    ("cloud 3000, from inside", cloud_scene(3000, 12.0, 7), CLOUD_CAMERA_INSIDE, 480, 270, 4),
    ("cloud 6000, from outside", cloud_scene(6000, 15.0, 11), CLOUD_CAMERA_OUTSIDE, 480, 270, 4),
]
tpt.InitializeTest()
for name, (s, m), cam, w, h, spp in SCENES:
    if only and only not in name:
        continue
    tpt.set_kernel_variant(hs, 3, -1)
    tpt.set_scene(s, m)
    tpt.set_camera(**cam)
    tpt.set_samples_per_pixel(spp)
    info = tpt.scene_info()
    tile = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    r0 = tpt.ray_counter_read()
    for f in range(frames):
        tpt.UpdateTest(0.0, f, w, h, 2)
        tpt.draw_device(0.0, f, w, h, tile.data_ptr(), 2)
        if f % 8 == 7 or f == frames - 1:
            tpt.synchronize()
            print("%-26s groups %5d frame %3d rays %12d image %08x" % (name, info["groups"], f, tpt.ray_counter_read() - r0, fnv1a(tile.cpu().numpy())), flush=True)
tpt.ShutdownTest()
