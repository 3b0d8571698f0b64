import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, torch
from toypathtracer_amd import api
from oracle_lib import Oracle, SEED_PER_PIXEL
o = Oracle.get()
api.InitializeTest()
api.set_kernel_variant(0, 3, -1)
for (w, h, spp, frames) in [(64, 64, 1, 1), (320, 184, 4, 3), (203, 117, 4, 2), (1280, 720, 4, 2)]:
    api.set_samples_per_pixel(spp)
    bb = np.zeros((h, w, 4), np.float32); per = []
    t0 = time.time()
    for f in range(frames):
        api.UpdateTest(0.0, f, w, h, 2); per.append(api.DrawTest(0.0, f, w, h, bb, 2))
    dt = time.time() - t0
    s, m = o.default_scene(); cam = o.default_camera(w, h); bo = np.zeros((h, w, 4), np.float32); pero = []
    for f in range(frames):
        r, _ = o.render(s, m, cam, w, h, spp, f, seed_mode=SEED_PER_PIXEL, backbuffer=bo); pero.append(r)
    print(w, h, spp, frames, 'rays', per, pero, 'equal', bb.tobytes() == bo.tobytes(), 'maxdiff', float(np.abs(bb - bo).max()), '%.3fs' % dt, flush=True)
api.ShutdownTest()
