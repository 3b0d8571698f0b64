"""Profiling build: how the grouped traversal behaves on the stress scene (group visits / exact tests per intersection)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["TPT_LIB"] = os.path.join(ROOT, "tools", "_stats", "libtoypathtracer_hip.so")
import torch
from toypathtracer_amd import api
from toypathtracer_amd.scenes import STRESS_CAMERA, stress_scene
api.InitializeTest()
s, m = stress_scene(4096, 64)
api.set_scene(s, m); api.set_camera(**STRESS_CAMERA); api.set_samples_per_pixel(8)
w, h = 1920, 1080
tile = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
api.UpdateTest(0.0, 0, w, h, 2)
api.debug_stats(True)
r0 = api.ray_counter_read()
api.draw_device(0.0, 0, w, h, tile.data_ptr(), 2)
rays = api.ray_counter_read() - r0
st = api.debug_stats(True)
steps, lanes = int(st[0]), int(st[32])
print("rays %d  intersection steps %d (%.1f lanes)" % (rays, steps, lanes / steps))
print("group visits: %.2f wave-trips per step, %.2f per ray (lanes per trip %.1f)" % (int(st[16]) / steps, int(st[48]) / rays, int(st[48]) / max(int(st[16]), 1)))
print("exact tests : %.2f wave-trips per step, %.2f per ray (lanes per trip %.1f)" % (int(st[1]) / steps, int(st[33]) / rays, int(st[33]) / max(int(st[1]), 1)))
api.ShutdownTest()
