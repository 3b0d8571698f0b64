"""Profiling build (TPT_LIB=tools/_stats/...): lane utilisation, queue fill and section shares over a pipelined BURST of frames."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["TPT_LIB"] = os.path.join(ROOT, "tools", "_stats", "libtoypathtracer_hip.so")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import time
import torch
from toypathtracer_amd import api
api.InitializeTest()
w, h, n = 1280, 720, int(os.environ.get("N", "20"))
tile = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
for rep in range(2):
    api.synchronize()
    api.debug_stats(True)
    r0 = api.ray_counter_read()
    t0 = time.perf_counter()
    for f in range(n):
        api.UpdateTest(0.0, f, w, h, 2)
        api.draw_device(0.0, f, w, h, tile.data_ptr(), 2)
    rays = api.ray_counter_read() - r0
    dt = time.perf_counter() - t0
    st = api.debug_stats(True)
    print("rep %d: %d frames %.2f ms  %.1f Mray/s  grid %d" % (rep, n, dt * 1e3, rays / dt / 1e6, api.launch_info()["grid_blocks"]))
    steps_w, steps_l = int(st[0]), int(st[32])
    print("  wave-steps %d  lane utilisation %.3f  ideal steps %d  idle polls %d" % (steps_w, steps_l / (64.0 * steps_w), rays // 64, int(st[12])))
    QN = ["FREE", "INT", "END", "DIEL", "METAL", "LAMBERT"]
    print("  fill:", "  ".join("%s %.1f (%d)" % (q, st[48 + c] / max(st[16 + c], 1), st[16 + c]) for c, q in enumerate(QN)))
    tot = float(sum(int(st[64 + k]) for k in range(25))) or 1.0
    print("  wave time: idle %.1f %%" % (100.0 * int(st[64 + 24]) / tot), "  ".join(
        "%s %.1f/%.1f/%.1f/%.1f" % ((q,) + tuple(100.0 * int(st[64 + c * 4 + k]) / tot for k in range(4))) for c, q in enumerate(QN)))
    waves = int(st[27])
api.ShutdownTest()
