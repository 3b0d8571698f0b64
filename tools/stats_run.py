"""Block-entry statistics of the trace kernel (profiling build, TPT_LIB=tools/_stats/libtoypathtracer_hip.so)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["TPT_LIB"] = os.path.join(ROOT, "tools", "_stats", "libtoypathtracer_hip.so")
import numpy as np
import torch
from toypathtracer_amd import api
NAMES = ["step", "phase2", "camera", "shadow", "sky", "hit", "lambert", "metal", "dielectric", "lightgen", "bounce", "finish",
         "refill", "chunk", "pixeldone", "diskloop", "sphereloop"]
api.InitializeTest()
for (w, h, spp, persist) in [(1280, 720, 4, 3), (3840, 2160, 16, 3)]:
    api.set_samples_per_pixel(spp)
    api.set_kernel_variant(0, persist, -1)
    import torch
    tile = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    api.UpdateTest(0.0, 0, w, h, 2)
    api.debug_stats(True)
    r0 = api.ray_counter_read()
    api.draw_device(0.0, 0, w, h, tile.data_ptr(), 2)
    rays = api.ray_counter_read() - r0
    st = api.debug_stats(True)
    print("== %dx%dx%d persist=%d rays=%d" % (w, h, spp, persist, rays))
    steps_w, steps_l = int(st[0]), int(st[32])
    print("  wave-steps %d lane-steps %d  lane utilisation %.3f  (rays/64 = %d)" % (steps_w, steps_l, steps_l / (64.0 * steps_w), rays // 64))
    waves = int(st[27])
    if waves:
        span = (int(st[26]) - int(st[25])) * 10e-9
        print("  waves %d  kernel span %.3f ms  mean wave lifetime %.3f ms  longest %.3f ms" % (waves, span * 1e3, int(st[24]) / waves * 10e-6, int(st[28]) * 10e-6))
    QN = ["FREE", "INT", "END", "DIEL", "METAL", "LAMBERT"]
    for c, n in enumerate(QN):
        if st[16 + c]:
            print("  queue %-8s batches %9d  paths %11d  fill %.1f / 64" % (n, st[16 + c], st[48 + c], st[48 + c] / st[16 + c]))
    tot = float(sum(int(st[64 + k]) for k in range(25))) or 1.0
    if tot > 1:
        print("  wave time by section (s_memtime ticks, share of all wave time): idle polls %.1f %%" % (100.0 * int(st[64 + 24]) / tot))
        for c, n in enumerate(QN):
            v = [int(st[64 + c * 4 + k]) for k in range(4)]
            print("    %-8s pick+pop %5.1f %%  class code %5.1f %%  intersect+classify %5.1f %%  push %5.1f %%" % ((n,) + tuple(100.0 * x / tot for x in v)))
    for i, n in enumerate(NAMES):
        if st[i]:
            print("  %-10s wave-entries %10d (%.3f per step)  lanes %12d (%.2f per entry)" % (n, st[i], st[i] / steps_w, st[32 + i], st[32 + i] / st[i]))
api.ShutdownTest()
