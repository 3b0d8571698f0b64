"""Light profiling build (tools/build_stats2.sh): claim / transition / idle counters and wave lifetimes over a pipelined burst."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TPT_LIB", os.path.join(ROOT, "tools", "_stats2", "libtoypathtracer_hip.so"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
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
    st = [int(x) for x in api.debug_stats(True)]
    print("rep %d: %d frames %.2f ms  %.1f Mray/s  grid %d" % (rep, n, dt * 1e3, rays / dt / 1e6, api.launch_info()["grid_blocks"]))
    print("  idle polls %d" % st[105])
    waves = max(st[109], 1)
    print("  waves %d  mean lifetime %.3f ms  steps/wave %.0f  lane utilisation %.3f  batches/wave %.0f  wave-time per step %.2f us" % (
        st[109], st[108] / waves * 1e-5, st[106] / waves, st[107] / max(64.0 * st[106], 1), st[110] / waves, st[108] * 0.01 / max(st[106], 1)))
    tot = float(sum(st[112:117])) or 1.0
    tot = float(sum(st[112:117])) or 1.0
    print("  hitSpheres: phase 1 %.1f %% of wave time, phase 2 %.1f %%; phase-2 trips per step %.2f, lanes busy per trip %.1f" % (
        100.0 * st[120] / tot, 100.0 * st[121] / tot, st[122] / max(st[106], 1), st[123] / max(st[122], 1)))
    print("  wave time: pick+pop %.1f %%  class code %.1f %%  intersect %.1f %%  push %.1f %%  idle %.1f %%" % tuple(100.0 * st[112 + k] / tot for k in range(5)))
api.ShutdownTest()
