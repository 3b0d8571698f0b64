import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["TPT_LIB"] = os.path.join(ROOT, "tools", "_stats", "libtoypathtracer_hip.so")
import numpy as np, torch
from toypathtracer_amd import api
api.InitializeTest()
w, h = 1280, 720
api.set_frame_overlap(1)
tile = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
for f in range(8):
    api.UpdateTest(0.0, f, w, h, 2)
    api.debug_stats(True)
    api.draw_device(0.0, f, w, h, tile.data_ptr(), 2)
    st = api.debug_stats(True)
    print("frame %d cost_order=%s wave-steps %d lane-steps %d util %.3f longest wave %.3f ms span %.3f ms" % (
        f, os.environ.get("TPT_COST_ORDER", "1"), st[0], st[32], st[32] / (64.0 * st[0]), int(st[28]) * 10e-6, (int(st[26]) - int(st[25])) * 10e-6))
api.ShutdownTest()
