"""CPU model of the persistent-wave scheduler (chunk queue + per-lane refill) on real per-pixel ray counts."""
import ctypes as C, sys, os, heapq
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle
emu = C.CDLL(os.path.join(ROOT, "tests/_build/liblane_emu.so"))
emu.emu_render_ex.restype = C.c_int64
emu.emu_render_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 6 + [C.c_uint] + [C.c_int] * 3 + [C.c_void_p, C.c_void_p]
o = Oracle.get()
w, h, spp = 1280, 720, int(sys.argv[1]) if len(sys.argv) > 1 else 4
s, m = o.default_scene(); cam = o.default_camera(w, h)
bb = np.zeros((h, w, 4), np.float32); pp = np.zeros((h, w), np.int32)
rays = emu.emu_render_ex(s.ctypes.data, m.ctypes.data, 46, cam.ctypes.data, w, h, 0, h, spp, 0, 2, 1, 0, 0, bb.ctypes.data, pp.ctypes.data)
print("rays", rays, "mean/pixel", pp.mean(), "max", pp.max(), "p99", np.percentile(pp, 99))
print("row means (bottom..top, every 60 rows):", [round(float(pp[y].mean()), 1) for y in range(0, h, 60)])
# tile-linear order
tiles = pp.reshape(h // 8, 8, w // 8, 8).transpose(0, 2, 1, 3).reshape(-1)  # idx -> rays
def simulate(nwaves, chunk, order=None):
    items = tiles if order is None else tiles.reshape(-1, 64)[order].reshape(-1)
    nchunks = len(items) // chunk
    # event-driven: each wave advances in lock-step; simulate wave by wave greedy by time
    # state per wave: remaining steps of its 64 lanes, current time (steps)
    next_chunk = 0
    waves = [(0, i) for i in range(nwaves)]  # (time, id)
    lanes = [np.zeros(64, np.int64) for _ in range(nwaves)]
    pool = [np.zeros(0, np.int64) for _ in range(nwaves)]
    heapq.heapify(waves)
    busy = 0; total_steps = 0; end = 0
    while waves:
        t, i = heapq.heappop(waves)
        L = lanes[i]
        # refill
        idle = np.where(L == 0)[0]
        k = 0
        while k < len(idle):
            if len(pool[i]) == 0:
                if next_chunk >= nchunks: break
                pool[i] = items[next_chunk * chunk:(next_chunk + 1) * chunk].astype(np.int64); next_chunk += 1
            n = min(len(idle) - k, len(pool[i]))
            L[idle[k:k + n]] = pool[i][:n]; pool[i] = pool[i][n:]; k += n
        act = L > 0
        if not act.any():
            end = max(end, t); continue
        # advance until the next lane finishes (all active lanes step together)
        d = int(L[act].min())
        busy += int(act.sum()) * d; total_steps += d
        L[act] -= d
        heapq.heappush(waves, (t + d, i))
    return busy / (64.0 * total_steps), total_steps, end
for nw, ch in [(3072, 64), (3072, 256), (2048, 64), (1024, 64), (4096, 64)]:
    u, steps, end = simulate(nw, ch)
    print("waves %d chunk %d: lane utilisation %.3f wave-steps %d makespan(steps) %d ideal %d" % (nw, ch, u, steps, end, tiles.sum() // 64 // nw))
# what would cost-sorted chunk order buy (longest-processing-time first, costs known from the previous frame)?
chunk_cost = tiles.reshape(-1, 64).sum(axis=1)
chunk_max = tiles.reshape(-1, 64).max(axis=1)
for name, order in [("by chunk sum desc", np.argsort(-chunk_cost)), ("by chunk max desc", np.argsort(-chunk_max))]:
    u, steps, end = simulate(4096, 64, order)
    print("%s: lane utilisation %.3f wave-steps %d makespan %d" % (name, u, steps, end))
u, steps, end = simulate(4096, 64)
print("image order (bottom-up): lane utilisation %.3f wave-steps %d makespan %d" % (u, steps, end))
# realistic version: order from the PREVIOUS frame's per-chunk statistics, applied to this frame
bb2 = np.zeros((h, w, 4), np.float32); pp2 = np.zeros((h, w), np.int32)
emu.emu_render_ex(s.ctypes.data, m.ctypes.data, 46, cam.ctypes.data, w, h, 0, h, spp, 1, 2, 1, 0, 0, bb2.ctypes.data, pp2.ctypes.data)
tiles_prev = tiles
tiles = pp2.reshape(h // 8, 8, w // 8, 8).transpose(0, 2, 1, 3).reshape(-1)
prev = tiles_prev.reshape(-1, 64)
for name, key in [("prev-frame chunk max", prev.max(axis=1)), ("prev-frame chunk sum", prev.sum(axis=1)),
                  ("prev-frame p90 within chunk", np.percentile(prev, 90, axis=1))]:
    u, steps, end = simulate(4096, 64, np.argsort(-key))
    print("frame 1 ordered by %s: lane utilisation %.3f wave-steps %d makespan %d" % (name, u, steps, end))
u, steps, end = simulate(4096, 64)
print("frame 1 image order: lane utilisation %.3f wave-steps %d makespan %d" % (u, steps, end))
