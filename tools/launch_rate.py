"""Per-frame pipeline period for frames with (almost) no work: what the command processor / runtime costs per frame."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from toypathtracer_amd import api
api.InitializeTest()
n = 3000
for (w, h, spp) in ((8, 8, 1), (64, 64, 1), (64, 64, 4)):
    api.set_samples_per_pixel(spp)
    tile = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    for persist in (3, 1):
        api.set_kernel_variant(0, persist, -1)
        line = "%dx%dx%d kernel %d:" % (w, h, spp, persist)
        for ov in (1, 2, 4, 8, 16):
            api.set_frame_overlap(ov)
            for f in range(100):
                api.UpdateTest(0.0, f, w, h, 2); api.draw_device(0.0, f, w, h, tile.data_ptr(), 2)
            api.synchronize()
            t0 = time.perf_counter()
            for f in range(n):
                api.UpdateTest(0.0, f, w, h, 2); api.draw_device(0.0, f, w, h, tile.data_ptr(), 2)
            api.synchronize()
            line += "  ov%d %.1f us" % (ov, (time.perf_counter() - t0) / n * 1e6)
        print(line, flush=True)
api.ShutdownTest()
