"""Soak test of the path-queue kernel: many pipelined frames, several sizes / spp / overlaps; every run must give the same
tile and ray count as the lane-refill kernel (different scheduling, same arithmetic) and as itself run again."""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from toypathtracer_amd import api
api.InitializeTest()

def run(w, h, spp, frames, persist, overlap, animate=False):
    api.set_samples_per_pixel(spp); api.set_kernel_variant(0, persist, -1); api.set_frame_overlap(overlap)
    tile = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    r0 = api.ray_counter_read()
    fl = 3 if animate else 2
    for f in range(frames):
        t = f * 0.05 if animate else 0.0
        api.UpdateTest(t, f, w, h, fl); api.draw_device(t, f, w, h, tile.data_ptr(), fl)
    rays = api.ray_counter_read() - r0
    return rays, zlib.crc32(tile.cpu().numpy().tobytes())

bad = 0
t0 = time.time()
SCALE = int(os.environ.get("SOAK_SCALE", "1"))  # multiply the frame counts
for (w, h, spp, frames) in [(1280, 720, 4, 150), (640, 360, 1, 300), (203, 117, 3, 400), (64, 8, 16, 400), (1920, 1080, 2, 40), (333, 5, 2, 300)]:
    frames *= SCALE
    ref = run(w, h, spp, frames, 1, 8)
    for ov in (16, 16, 5, 1):
        got = run(w, h, spp, frames if ov > 1 else min(frames, 60), 3, ov)
        exp = ref if ov > 1 or frames <= 60 else run(w, h, spp, 60, 1, 8)
        ok = got == exp
        bad += not ok
        print("%4dx%-4d spp %2d frames %3d overlap %2d: rays %d crc %08x %s" % (w, h, spp, frames, ov, got[0], got[1], "ok" if ok else "MISMATCH vs %r" % (exp,)), flush=True)
# row-sharded: the union of the parts' tiles over many pipelined frames == the unsharded tile
import numpy as np
def run_sharded(w, h, spp, frames, parts, stripe, persist, overlap):
    api.set_samples_per_pixel(spp); api.set_kernel_variant(0, persist, -1); api.set_frame_overlap(overlap)
    full = np.zeros((h, w, 4), np.float32); rays = 0
    for p in range(parts):
        api.set_row_shard(stripe, parts, p)
        rows = api.local_row_count(h)
        tile = torch.zeros((max(rows, 1), w, 4), dtype=torch.float32, device="cuda")
        r0 = api.ray_counter_read()
        for f in range(frames):
            api.UpdateTest(0.0, f, w, h, 2); api.draw_device(0.0, f, w, h, tile.data_ptr(), 2)
        rays += api.ray_counter_read() - r0
        t = tile.cpu().numpy()
        for ly in range(rows):
            full[api.local_row_to_global(ly)] = t[ly]
    api.set_row_shard(0, 1, 0)
    return rays, zlib.crc32(full.tobytes())
for (w, h, spp, frames, parts, stripe) in [(1280, 720, 4, 100, 8, 8), (640, 360, 2, 150, 3, 16), (203, 117, 4, 200, 5, 4)]:
    exp = run(w, h, spp, frames, 3, 16)
    for persist, ov in ((3, 8), (3, 16), (1, 8)):
        got = run_sharded(w, h, spp, frames, parts, stripe, persist, ov)
        ok = got == exp; bad += not ok
        print("%4dx%-4d spp %2d frames %3d sharded %d x %d rows, kernel %d overlap %2d: %s" % (w, h, spp, frames, parts, stripe, persist, ov, "ok" if ok else "MISMATCH %r vs %r" % (got, exp)), flush=True)
# stress scene, a few pipelined frames, both kernels
from toypathtracer_amd.scenes import STRESS_CAMERA, stress_scene
ss, sm = stress_scene(4096, 64)
api.set_scene(ss, sm); api.set_camera(**STRESS_CAMERA)
a = run(480, 270, 2, 24, 1, 8); b = run(480, 270, 2, 24, 3, 16)
print("stress 4096 spheres 480x270: %s" % ("ok" if a == b else "MISMATCH %r %r" % (a, b))); bad += a != b
api.set_scene(None); api.set_camera(None)
a = run(640, 360, 4, 120, 1, 8, True); b = run(640, 360, 4, 120, 3, 16, True)
print("animated 640x360: %s" % ("ok" if a == b else "MISMATCH"))
bad += a != b
print("soak done in %.1f s, mismatches: %d" % (time.time() - t0, bad))
api.ShutdownTest()
sys.exit(1 if bad else 0)
