"""Hunt for a rare mismatch: the frame-overlap test's scenario (192x128, 19 frames, overlap 8, kernel timing on), hundreds of
times, with the transitions the test suite makes around it (overlap 1/2/3/16 runs in between)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from toypathtracer_amd import api
from oracle_lib import Oracle, SEED_PER_PIXEL
o = Oracle.get()
api.InitializeTest()
w, h = 192, 128
s, m = o.default_scene(); cam = o.default_camera(w, h)
ref = {}
def oracle_frames(frames):
    if frames not in ref:
        bo = np.zeros((h, w, 4), np.float32); ro = 0
        for f in range(frames):
            r, _ = o.render(s, m, cam, w, h, 4, f, seed_mode=SEED_PER_PIXEL, backbuffer=bo); ro += r
        ref[frames] = (ro, bo.copy())
    return ref[frames]
bad = 0
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
if os.environ.get("FLAKE_REINIT"):  # what tests/test_gpu_api.py::test_shutdown_and_reinitialise does before the other suites run
    for cycle in range(int(os.environ["FLAKE_REINIT"])):
        bb = np.zeros((64, 96, 4), np.float32)
        for f in range(2):
            api.UpdateTest(0.0, f, 96, 64, 2); api.DrawTest(0.0, f, 96, 64, bb, 2)
        api.ShutdownTest(); api.InitializeTest()
t0 = time.time()
for it in range(iters):
    for ov in (1, 2, 3, 8, 16):
        frames = 7 if ov < 8 else 2 * ov + 3
        api.set_frame_overlap(ov)
        tile = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
        if it % 2: torch.cuda.synchronize()          # odd iterations: rule out the zero-fill racing the first blend
        r0 = api.ray_counter_read()
        timing = (it % 4) < 2
        if timing: api.kernel_timing_begin(frames)
        for f in range(frames):
            api.UpdateTest(0.0, f, w, h, 2); api.draw_device(0.0, f, w, h, tile.data_ptr(), 2)
        if timing: ms, n = api.kernel_timing_end()
        rays = api.ray_counter_read() - r0
        got = tile.cpu().numpy()
        ro, bo = oracle_frames(frames)
        if rays != ro or got.tobytes() != bo.tobytes() or (timing and (n != frames or not ms > 0)):
            bad += 1
            diff = np.argwhere((got != bo).any(axis=2))
            print("MISMATCH it %d overlap %d timing %s sync %d: rays %d vs %d, %d pixels differ (first %s), timing n=%s ms=%s" % (
                it, ov, timing, it % 2, rays, ro, len(diff), diff[:3].tolist(), n if timing else None, ms if timing else None), flush=True)
print("flake hunt: %d iterations x 5 overlaps in %.1f s, mismatches %d" % (iters, time.time() - t0, bad))
api.ShutdownTest()
