"""Does a freshly uploaded scene set ever reach a trace kernel on another stream stale?  Alternates two different scenes of
the same size (so every one of the 32 scene sets' blobs keeps being overwritten with different data), a few pipelined
frames each, against the oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from toypathtracer_amd import api
from oracle_lib import Oracle, SEED_PER_PIXEL
o = Oracle.get()
api.InitializeTest()
w, h, frames = 192, 128, 6
sA, mA = o.default_scene()
sB, mB = sA.copy(), mA.copy()
sB["cx"] += 0.37; sB["cz"] -= 0.21; sB["radius"][1:] *= 0.8; sB["invRadius"] = 1.0 / sB["radius"]
cam = o.default_camera(w, h)
ref = []
for s_, m_ in ((sA, mA), (sB, mB)):
    bo = np.zeros((h, w, 4), np.float32); ro = 0
    for f in range(frames):
        r, _ = o.render(s_, m_, cam, w, h, 4, f, seed_mode=SEED_PER_PIXEL, backbuffer=bo); ro += r
    ref.append((ro, bo.copy()))
bad = 0
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for it in range(iters):
    k = it & 1
    api.set_scene(*((sA, mA), (sB, mB))[k])
    api.set_frame_overlap(8 if it % 3 else 16)
    tile = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
    r0 = api.ray_counter_read()
    for f in range(frames):
        api.UpdateTest(0.0, f, w, h, 2); api.draw_device(0.0, f, w, h, tile.data_ptr(), 2)
    rays = api.ray_counter_read() - r0
    got = tile.cpu().numpy()
    if rays != ref[k][0] or got.tobytes() != ref[k][1].tobytes():
        bad += 1
        print("MISMATCH it %d scene %d: rays %d vs %d, %d pixels differ" % (it, k, rays, ref[k][0], int((got != ref[k][1]).any(axis=2).sum())), flush=True)
print("restage hunt: %d iterations, mismatches %d" % (iters, bad))
api.ShutdownTest()
