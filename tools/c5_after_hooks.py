"""The suite's order around the C5 first-render pixel difference (round 5): large frames (slot buffers grow), small frames (they shrink),
the HOOKS build of the library used and -- new in round 5 -- shut down again in the same process, then the 4096-sphere frame through
DrawTest twice.  Modes: hooks (use + shut down), keep (use, keep alive as round 4 did), none.
    python tools/c5_after_hooks.py [reps] [mode]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle_lib import fnv1a  # noqa: E402
from toypathtracer_amd import api as tpt  # noqa: E402
from toypathtracer_amd.scenes import STRESS_CAMERA, stress_scene  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
mode = sys.argv[2] if len(sys.argv) > 2 else "hooks"
if os.environ.get("C5_BYPASS_GUARD"):  # HIP starts with the queues the environment asks for; the library's rule (> 22 -> bounds on the VALU) sees 20
    torch.cuda.init(); torch.zeros(1, device="cuda")
    os.environ["GPU_MAX_HW_QUEUES"] = "20"
tpt.InitializeTest()
s, m = stress_scene(4096, int(os.environ.get("C5_LIGHTS", "64")))
seen = {}
first = {}
extra_streams = []
for r in range(reps):
    tpt.set_scene(None); tpt.set_camera(None); tpt.set_samples_per_pixel(4)
    big = torch.zeros((2160, 3840, 4), dtype=torch.float32, device="cuda")
    tpt.UpdateTest(0.0, 0, 3840, 2160, 2); tpt.draw_device(0.0, 0, 3840, 2160, big.data_ptr(), 2); tpt.synchronize()   # slots grow to 132 MB
    del big
    small = np.zeros((117, 203, 4), np.float32)
    for f in range(12):                                                                                                    # ... and shrink again
        tpt.UpdateTest(0.0, f, 203, 117, 2); tpt.DrawTest(0.0, f, 203, 117, small, 2)
    disturb = os.environ.get("C5_DISTURB", "hooks_lane")  # what the second context does: hooks_lane (20 frames on the lane-refill kernel),
    #                                                        hooks_queue (20 frames on the shipped kernel), hooks_init (nothing), torch (no second
    #                                                        context at all: a matrix product by torch), hooks_lane_sleep (as hooks_lane, then idle 0.3 s)
    if mode != "none" and disturb == "torch_streams":  # no second context: just more streams (= hardware queues) in the process
        if not extra_streams:
            for _ in range(int(os.environ.get("C5_TORCH_STREAMS", "16"))):
                st_ = torch.cuda.Stream()
                with torch.cuda.stream(st_):
                    torch.zeros(16, device="cuda").add_(1.0)
                extra_streams.append(st_)
            torch.cuda.synchronize()
    elif mode != "none" and disturb == "torch":
        x = torch.randn((2048, 2048), device="cuda")
        for _ in range(4):
            x = (x @ x) * 1e-3
        torch.cuda.synchronize()
    elif mode != "none":
        cm = tpt.using_hooks()
        cm.__enter__()
        if disturb != "hooks_init":
            if disturb != "hooks_queue":
                tpt.set_kernel_variant(0, 1, -1)
            tile = torch.zeros((360, 640, 4), dtype=torch.float32, device="cuda")
            for f in range(20):
                tpt.UpdateTest(0.0, f, 640, 360, 2); tpt.draw_device(0.0, f, 640, 360, tile.data_ptr(), 2)
            tpt.ray_counter_read()
            tpt.synchronize()
            if disturb == "hooks_lane_sleep":
                torch.cuda.synchronize()
                import time
                time.sleep(0.3)
        if mode == "hooks":
            cm.__exit__(None, None, None)      # shuts the hooks context down (round 5)
        else:
            tpt._lib = cm._prev                # round 4: back to the product, the hooks context stays alive
    W, H = 1920, 1080
    if os.environ.get("C5_SCENE", "stress") == "default4k":   # the 9-sphere scene (the kernel that stages the scene in LDS), launches of ~50 ms
        tpt.set_scene(None); tpt.set_camera(None); tpt.set_samples_per_pixel(int(os.environ.get("C5_SPP", "64")))
        W, H = 3840, 2160
    else:
        tpt.set_scene(s, m); tpt.set_camera(**STRESS_CAMERA); tpt.set_samples_per_pixel(8)
        if "C5_VARIANT" in os.environ:   # e.g. "3,3,-1": the shipped kernel with the packed VALU filter for the groups' bounds (no matrix cores)
            tpt.set_kernel_variant(*[int(v) for v in os.environ["C5_VARIANT"].split(",")])
    out = []
    path = os.environ.get("C5_PATH", "drawtest")  # drawtest: host buffer (look-ahead frames behind it); device: tptDrawDevice + synchronise
    if "C5_LOOKAHEAD" in os.environ:
        tpt.set_host_lookahead(int(os.environ["C5_LOOKAHEAD"]))
    for k in range(2):
        if path == "drawtest":
            bb = np.zeros((H, W, 4), np.float32)
            tpt.UpdateTest(0.0, 0, W, H, 2)
            out.append((tpt.DrawTest(0.0, 0, W, H, bb, 2), "%08x" % fnv1a(bb)))
        else:
            n = int(os.environ.get("C5_INFLIGHT", "3"))
            tiles = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(n)]
            r0 = tpt.ray_counter_read()
            for f in range(n):
                tpt.UpdateTest(0.0, f, W, H, 2)
                tpt.draw_device(0.0, f, W, H, tiles[f].data_ptr(), 2)
            tpt.synchronize()
            imgs = [t.cpu().numpy() for t in tiles]
            out.append((tpt.ray_counter_read() - r0, " ".join("%08x" % fnv1a(i) for i in imgs)))
            for f, i in enumerate(imgs):  # a frame whose image differs from the first one seen: which pixels, which values
                ref = first.setdefault(f, i)
                bad = (ref != i).any(axis=2)
                for y, x in list(zip(*np.nonzero(bad)))[:3]:
                    print("   frame %d pixel (%d, %d): %s instead of %s (%d pixels differ)" % (f, x, y, i[y, x, :3], ref[y, x, :3], bad.sum()))
    for key in out:
        seen[key] = seen.get(key, 0) + 1
    print("rep %d: %s" % (r, out), flush=True)
print("mode %s results: %s" % (mode, seen))
tpt.ShutdownTest()
