"""Light profiling build (tools/build_stats2.sh) on the 4096-sphere frame: wave time by section, and inside the intersections the stages of
the grouped dealing (tpt_kernels.hip TPT_DEAL_T: big spheres / super-groups' bounds / groups' bounds + list / member filter / exact tests)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TPT_LIB", os.path.join(ROOT, "tools", "_stats2", "libtoypathtracer_hip.so"))
import torch
from toypathtracer_amd import api
from toypathtracer_amd.scenes import STRESS_CAMERA, stress_scene
api.InitializeTest()
s, m = stress_scene(4096, 64)
api.set_scene(s, m); api.set_camera(**STRESS_CAMERA); api.set_samples_per_pixel(8)
w, h, n = 1920, 1080, int(os.environ.get("N", "3"))
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
    st = [int(x) for x in api.debug_stats(True)]
    print("rep %d: %d frames %.2f ms  %.1f Mray/s  %s" % (rep, n, dt * 1e3, rays / dt / 1e6, api.launch_info()))
    waves, steps = max(st[109], 1), max(st[106], 1)
    print("  waves %d  mean lifetime %.3f ms  steps/wave %.0f  lane utilisation %.3f  wave-time per step %.2f us" % (
        st[109], st[108] / waves * 1e-5, steps / waves, st[107] / (64.0 * steps), st[108] * 0.01 / steps))
    tot = float(sum(st[112:117])) or 1.0
    print("  wave time: pick+pop %.1f %%  class code %.1f %%  intersect %.1f %%  push %.1f %%  idle %.1f %%" % tuple(100.0 * st[112 + k] / tot for k in range(5)))
    calls = max(st[99], 1)
    names = ["big spheres", "super-groups' bounds (wave-wide)", "groups' bounds + group entries (stage B)", "member filter (list, parked ray, gathers)", "survivors dealt + exact tests"]
    for k, nm in enumerate(names):
        print("    %-44s %5.1f %% of wave time, %7.0f ticks per call" % (nm, 100.0 * st[90 + k] / tot, st[90 + k] / calls))
    print("    %-44s %5.1f %% of wave time, %7.0f ticks per call" % ("super-group entries written (stage A)", 100.0 * st[104] / tot, st[104] / calls))
    print("  per call (64 lanes): rounds %.2f, sub-rounds of 64 pairs %.2f, pairs %.1f, survivors %.1f, exact passes %.2f; rays per call %.1f" % (
        st[97] / calls, st[95] / calls, st[96] / calls, st[98] / calls, st[100] / calls, rays / calls))
    print("  wave trips per call: stage A entry loop %.2f, stage B entry loop %.2f, survivors' push loop %.2f" % (st[101] / calls, st[102] / calls, st[103] / calls))
    print("  behind the origin (centre behind, origin outside the bound): %.1f %% of the (ray, super-group) entries, %.1f %% of the (ray, group) candidates they produce" % (
        100.0 * st[124] / max(st[126], 1), 100.0 * st[125] / max(st[96], 1)))
api.ShutdownTest()
