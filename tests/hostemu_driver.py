"""TEST INFRASTRUCTURE: scenarios that drive csrc/tpt_host.cpp -- compiled against tests/hostemu's stand-in for the HIP runtime and
host restatements of the kernels -- through the same ctypes mirror the GPU tests use, and hold every image and ray count against the
oracle.  Run by tests/test_host_logic.py in a subprocess with TPT_LIB=tests/_build/libtpt_hostemu.so and HOSTEMU_POLICY=eager|lazy|
random:<seed>; prints one line per scenario.  The scenarios are the GPU suite's (tests/test_gpu_api.py, test_gpu_parity.py) at sizes a
CPU renders in a fraction of a second."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from common import oracle_frames  # noqa: E402
from oracle_lib import FLAG_ANIMATE, FLAG_PROGRESSIVE, SEED_PER_PIXEL, SEED_ROW_SERIAL, Oracle  # noqa: E402

assert "hostemu" in os.environ.get("TPT_LIB", ""), "this driver is for the host-emulation build only"
from toypathtracer_amd import api as tpt  # noqa: E402

o = Oracle.get()
W, H, SPP = 64, 40, 2


def ptr(a):
    return a.ctypes.data


def same(got, want, what):
    if got.tobytes() != want.tobytes():
        bad = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
        raise AssertionError("%s: %d words differ, first at %s" % (what, len(bad), bad[0] if len(bad) else None))


def reset():
    tpt.set_samples_per_pixel(SPP)
    tpt.set_seed_mode(SEED_PER_PIXEL)
    tpt.set_fold_mode(0)
    tpt.set_kernel_variant(0, 3, -1)
    tpt.set_config()
    tpt.set_scene()
    tpt.set_camera()
    tpt.set_frame_overlap(16)
    tpt.set_row_shard(0, 1, 0)
    tpt.set_host_lookahead(2)
    tpt.set_host_buffer_mode(0)
    tpt.set_stream_batching(1)
    tpt.set_tile_mirror(0)
    tpt.set_ray_counter(0)
    tpt.synchronize()


def streaming(frames=12, w=W, h=H, flags=FLAG_PROGRESSIVE, batching=True, **variant):
    """enqueue, enqueue, ..., one synchronise: image and ray total"""
    reset()
    tpt.set_stream_batching(1 if batching else 0)
    if variant:
        tpt.set_kernel_variant(**variant)
    tile = np.zeros((h, w, 4), np.float32)
    r0 = tpt.ray_counter_read()
    for f in range(frames):
        tpt.UpdateTest(f / 60.0, f, w, h, flags)
        tpt.draw_device(f / 60.0, f, w, h, ptr(tile), flags)
    tpt.synchronize()
    rays = tpt.ray_counter_read() - r0
    if flags & FLAG_ANIMATE:
        return  # (the animated scene is held against the oracle frame by frame in scenario `animated`)
    total, want, _ = oracle_frames(o, w, h, SPP, frames, seed_mode=SEED_PER_PIXEL)
    same(tile, want, "streaming %dx%d %s" % (w, h, variant))
    assert rays == total, (rays, total)


def synchronous_device_caller(frames=10):
    """a synchronise after every frame: from the third frame on the next ones are traced ahead"""
    reset()
    tile = np.zeros((H, W, 4), np.float32)
    total, want, per_frame = oracle_frames(o, W, H, SPP, frames, seed_mode=SEED_PER_PIXEL)
    hits0 = tpt.lookahead_hits()
    last = tpt.ray_counter_read()
    for f in range(frames):
        tpt.UpdateTest(0.0, f, W, H, FLAG_PROGRESSIVE)
        tpt.draw_device(0.0, f, W, H, ptr(tile), FLAG_PROGRESSIVE)
        tpt.synchronize()
        now = tpt.ray_counter_read()
        assert now - last == per_frame[f], (f, now - last, per_frame[f])
        last = now
    same(tile, want, "synchronous device caller")
    return tpt.lookahead_hits() - hits0


def drawtest_host(frames=8, seed_mode=SEED_PER_PIXEL, lookahead=2, w=W, h=H):
    """the reference's own contract: DrawTest(host float*) returns when the frame is in the buffer, with its ray count"""
    reset()
    tpt.set_seed_mode(seed_mode)
    tpt.set_host_lookahead(lookahead)
    bb = np.zeros((h, w, 4), np.float32)
    total, want, per_frame = oracle_frames(o, w, h, SPP, frames, seed_mode=seed_mode)
    for f in range(frames):
        tpt.UpdateTest(0.0, f, w, h, FLAG_PROGRESSIVE)
        rays = tpt.DrawTest(0.0, f, w, h, bb, FLAG_PROGRESSIVE)
        assert rays == per_frame[f], (f, rays, per_frame[f])
    same(bb, want, "DrawTest seed mode %d lookahead %d" % (seed_mode, lookahead))


def lookahead_dropped_by_state_changes():
    """frames traced ahead on a guess are thrown away when anything they depend on changes"""
    reset()
    bb = np.zeros((H, W, 4), np.float32)
    for f in range(4):
        tpt.UpdateTest(0.0, f, W, H, FLAG_PROGRESSIVE)
        tpt.DrawTest(0.0, f, W, H, bb, FLAG_PROGRESSIVE)
    tpt.set_samples_per_pixel(1)  # the frames traced ahead used 2 spp
    s, m = o.default_scene()
    cam = o.default_camera(W, H)
    want = bb.copy()
    for f in range(4, 7):
        tpt.UpdateTest(0.0, f, W, H, FLAG_PROGRESSIVE)
        rays = tpt.DrawTest(0.0, f, W, H, bb, FLAG_PROGRESSIVE)
        r, _ = o.render(s, m, cam, W, H, 1, f, FLAG_PROGRESSIVE, backbuffer=want, seed_mode=SEED_PER_PIXEL)
        assert rays == r, (f, rays, r)
    same(bb, want, "look-ahead after a change of spp")
    # a repeated frame number is not "the next frame" either
    tpt.UpdateTest(0.0, 6, W, H, FLAG_PROGRESSIVE)
    rays = tpt.DrawTest(0.0, 6, W, H, bb, FLAG_PROGRESSIVE)
    r, _ = o.render(s, m, cam, W, H, 1, 6, FLAG_PROGRESSIVE, backbuffer=want, seed_mode=SEED_PER_PIXEL)
    assert rays == r
    same(bb, want, "look-ahead after a repeated frame")


def batches(sizes=(3, 1, 4), seed_mode=SEED_PER_PIXEL):
    """tptDrawDeviceBatch: the same bits as one call per frame"""
    reset()
    tpt.set_seed_mode(seed_mode)
    tile = np.zeros((H, W, 4), np.float32)
    f = 0
    r0 = tpt.ray_counter_read()
    for k in sizes:
        tpt.UpdateTest(0.0, f, W, H, FLAG_PROGRESSIVE)
        tpt.draw_device_batch(0.0, f, k, W, H, ptr(tile), FLAG_PROGRESSIVE)
        f += k
    tpt.synchronize()
    total, want, _ = oracle_frames(o, W, H, SPP, f, seed_mode=seed_mode)
    same(tile, want, "batches %s seed mode %d" % (sizes, seed_mode))
    assert tpt.ray_counter_read() - r0 == total


def stream_batching(frames=21, w=32, h=24):
    """small frames of a streaming caller are traced several per launch behind its back; every frame is still delivered"""
    reset()
    tile = np.zeros((h, w, 4), np.float32)
    total, want, per_frame = oracle_frames(o, w, h, SPP, frames, seed_mode=SEED_PER_PIXEL)
    r0 = tpt.ray_counter_read()
    for f in range(frames):
        tpt.UpdateTest(0.0, f, w, h, FLAG_PROGRESSIVE)
        tpt.draw_device(0.0, f, w, h, ptr(tile), FLAG_PROGRESSIVE)
    tpt.synchronize()
    same(tile, want, "stream batching")
    assert tpt.ray_counter_read() - r0 == total
    return tpt.launch_info()


def animated(frames=6):
    """kFlagAnimate: UpdateTest moves a sphere every frame; every frame needs its own scene set in flight"""
    reset()
    tile = np.zeros((H, W, 4), np.float32)
    flags = FLAG_PROGRESSIVE | FLAG_ANIMATE
    s, m = o.default_scene()
    cam = o.default_camera(W, H)
    want = np.zeros((H, W, 4), np.float32)
    total = 0
    r0 = tpt.ray_counter_read()
    for f in range(frames):
        t = f / 60.0
        tpt.UpdateTest(t, f, W, H, flags)
        tpt.draw_device(t, f, W, H, ptr(tile), flags)
        o.animate(s, t)
        r, _ = o.render(s, m, cam, W, H, SPP, f, flags, backbuffer=want, seed_mode=SEED_PER_PIXEL)
        total += r
    tpt.synchronize()
    same(tile, want, "animated scene")
    assert tpt.ray_counter_read() - r0 == total


def resizes():
    """frame shapes that grow, shrink and grow again while frames are in flight (buffer re-allocation drains the pipeline first)"""
    reset()
    for (w, h, frames) in [(32, 24, 5), (96, 64, 4), (40, 40, 6), (96, 64, 3)]:
        tile = np.zeros((h, w, 4), np.float32)
        for f in range(frames):
            tpt.UpdateTest(0.0, f, w, h, FLAG_PROGRESSIVE)
            tpt.draw_device(0.0, f, w, h, ptr(tile), FLAG_PROGRESSIVE)
        tpt.synchronize()
        _, want, _ = oracle_frames(o, w, h, SPP, frames, seed_mode=SEED_PER_PIXEL)
        same(tile, want, "resize to %dx%d" % (w, h))


def sharded_loopback(n=4, frames=10, w=80, h=56, stripe=8):
    """rank 0 of n on a loopback communicator: its stripes equal the 1-GPU render, the other ranks' rows stay zero"""
    reset()
    tpt.comm_init_loopback(n, stripe)
    try:
        img = np.zeros((h, w, 4), np.float32)
        for f in range(frames):
            tpt.UpdateTest(0.0, f, w, h, FLAG_PROGRESSIVE)
            tpt.draw_sharded(0.0, f, w, h, ptr(img), FLAG_PROGRESSIVE)
        tpt.sharded_finish()
    finally:
        tpt.comm_destroy()
    _, want, _ = oracle_frames(o, w, h, SPP, frames, seed_mode=SEED_PER_PIXEL)
    mine = (np.arange(h) // stripe) % n == 0
    same(img[mine], want[mine], "sharded loopback n=%d" % n)
    assert not img[~mine].any()


def sharded_batches(n=4, w=80, h=56, stripe=8, sizes=(3, 1, 4)):
    """tptDrawShardedBatch: rank 0's stripes after batches of 3 + 1 + 4 frames equal the 1-GPU render of 8 frames"""
    reset()
    tpt.comm_init_loopback(n, stripe)
    try:
        img = np.zeros((h, w, 4), np.float32)
        f = 0
        for k in sizes:
            tpt.UpdateTest(0.0, f, w, h, FLAG_PROGRESSIVE)
            tpt.draw_sharded_batch(0.0, f, k, w, h, ptr(img), FLAG_PROGRESSIVE)
            f += k
        tpt.sharded_finish()
    finally:
        tpt.comm_destroy()
    _, want, _ = oracle_frames(o, w, h, SPP, f, seed_mode=SEED_PER_PIXEL)
    mine = (np.arange(h) // stripe) % n == 0
    same(img[mine], want[mine], "sharded batches")
    assert not img[~mine].any()


def sharded_deferred_state_changes(n=4, w=80, h=56, stripe=8):
    """tptDrawSharded collects small tiles into batches (tptSetShardExchangeInterval, automatic: 4 frames here).  Frames that were
    accepted but not issued yet must be rendered with the configuration they were accepted under: a camera / scene / spp change, a
    change of the interval itself, a counter read and the closing tptShardedFinish each issue what is pending first."""
    reset()
    from toypathtracer_amd.scenes import stress_scene
    s0, m0 = o.default_scene()
    cam0 = o.default_camera(w, h)
    s1, m1 = stress_scene(300, 18)
    cam1 = o.camera((0, 3, 9), (0, 0, 0), (0, 1, 0), 50.0, w / h, 0.0, 9.0)
    want = np.zeros((h, w, 4), np.float32)
    img = np.zeros((h, w, 4), np.float32)
    total = 0
    tpt.comm_init_loopback(n, stripe)
    try:
        r0 = tpt.sharded_finish()
        s, m, cam, spp = s0, m0, cam0, SPP
        for f in range(17):
            if f == 2:   # two frames pending: the camera moves
                tpt.set_camera((0, 3, 9), (0, 0, 0), 50.0, 0.0, 9.0)
                cam = cam1
            if f == 5:   # three pending: another scene
                tpt.set_scene(s1, m1)
                s, m = s1, m1
            if f == 7:   # two pending: more samples per pixel
                tpt.set_samples_per_pixel(3)
                spp = 3
            if f == 10:  # three pending: the host's own interval
                tpt.set_shard_exchange_interval(3)
            if f == 14:  # one pending (10-12 went out as a batch of three): a counter read waits for everything accepted so far
                assert tpt.ray_counter_read() - r0 == total
            tpt.UpdateTest(0.0, f, w, h, FLAG_PROGRESSIVE)
            tpt.draw_sharded(0.0, f, w, h, ptr(img), FLAG_PROGRESSIVE)
            for y0 in range(0, h, stripe * n):  # rank 0's stripes
                r, _ = o.render(s, m, cam, w, h, spp, f, FLAG_PROGRESSIVE, backbuffer=want, seed_mode=SEED_PER_PIXEL, y0=y0, y1=min(y0 + stripe, h))
                total += r
        assert tpt.sharded_finish() - r0 == total  # (frames 14-16: two pending at the end)
    finally:
        tpt.set_shard_exchange_interval(0)
        tpt.comm_destroy()
    mine = (np.arange(h) // stripe) % n == 0
    same(img[mine], want[mine], "deferred sharded frames across state changes")
    assert not img[~mine].any()


def custom_scene_and_camera():
    """tptSetScene / tptSetCamera between frames of a stream: the frames before see the old scene, the ones after the new one"""
    reset()
    from toypathtracer_amd.scenes import stress_scene
    tile = np.zeros((H, W, 4), np.float32)
    want = np.zeros((H, W, 4), np.float32)
    s0, m0 = o.default_scene()
    cam0 = o.default_camera(W, H)
    s1, m1 = stress_scene(300, 18)
    cam1 = o.camera((0, 3, 9), (0, 0, 0), (0, 1, 0), 50.0, W / H, 0.0, 9.0)
    for f in range(6):
        if f == 3:
            tpt.set_scene(s1, m1)
            tpt.set_camera((0, 3, 9), (0, 0, 0), 50.0, 0.0, 9.0)
        tpt.UpdateTest(0.0, f, W, H, FLAG_PROGRESSIVE)
        tpt.draw_device(0.0, f, W, H, ptr(tile), FLAG_PROGRESSIVE)
        s, m, cam = (s0, m0, cam0) if f < 3 else (s1, m1, cam1)
        o.render(s, m, cam, W, H, SPP, f, FLAG_PROGRESSIVE, backbuffer=want, seed_mode=SEED_PER_PIXEL)
    tpt.synchronize()
    same(tile, want, "scene and camera change in a stream")


def display_counter_and_mirror():
    """tptDisplayRGBA8 (Cpp/Emscripten/main.cpp:63-79), a caller-owned ray counter, the tile mirror with its counter snapshot"""
    reset()
    tile = np.zeros((H, W, 4), np.float32)
    mirror = np.zeros((H, W, 4), np.float32)
    counter = np.zeros(1, np.int64)
    snapshot = np.zeros(2, np.int64)
    tpt.set_ray_counter(ptr(counter))
    tpt.set_stream_batching(0)
    total, want, _ = oracle_frames(o, W, H, SPP, 5, seed_mode=SEED_PER_PIXEL)
    for f in range(5):
        tpt.UpdateTest(0.0, f, W, H, FLAG_PROGRESSIVE)
        tpt.set_tile_mirror(ptr(mirror), ptr(snapshot))
        tpt.draw_device(0.0, f, W, H, ptr(tile), FLAG_PROGRESSIVE)
    rgba = np.zeros((H, W, 4), np.uint8)
    tpt.display_rgba8(ptr(tile), W, H, ptr(rgba))
    tpt.synchronize()
    same(tile, want, "tile")
    same(mirror, want, "mirror of the tile")
    assert counter[0] == total and snapshot[0] == total, (counter[0], snapshot[0], total)
    ref = np.empty((H, W, 4), np.uint8)
    ref[..., :3] = np.minimum(np.sqrt(want[::-1, :, :3]) * np.float32(255), np.float32(255.0)).astype(np.uint8)
    ref[..., 3] = 255
    assert np.array_equal(rgba, ref)
    tpt.set_tile_mirror(0)
    tpt.set_ray_counter(0)


def errors_and_reinitialisation():
    """bad arguments fail with a message and change nothing; draw before update fails; shutdown + initialise starts clean"""
    reset()
    tile = np.zeros((H, W, 4), np.float32)
    for bad in (lambda: tpt.draw_device(0.0, 0, 0, H, ptr(tile), FLAG_PROGRESSIVE), lambda: tpt.draw_device(0.0, 0, W, H, 0, FLAG_PROGRESSIVE),
                lambda: tpt.set_samples_per_pixel(0), lambda: tpt.set_kernel_variant(7, 3, -1), lambda: tpt.draw_device_batch(0.0, 0, 0, W, H, ptr(tile), FLAG_PROGRESSIVE)):
        try:
            bad()
        except tpt.TptError:
            continue
        raise AssertionError("a bad call was accepted")
    tpt.ShutdownTest()
    tpt.InitializeTest()
    tpt.set_samples_per_pixel(SPP)
    try:
        tpt.draw_device(0.0, 0, W, H, ptr(tile), FLAG_PROGRESSIVE)
        raise AssertionError("draw before update was accepted")
    except tpt.TptError:
        pass
    # the reference's own void DrawTest (Test.h:14) has no error channel: with a handler installed (tptSetErrorHandler) the failure is
    # handed to it and the call returns without effect -- without one the library would print and abort()
    seen = []
    tpt.set_error_handler(lambda where, msg: seen.append((where, msg)))
    try:
        lib = tpt.load_library()
        fn = getattr(lib, "_Z8DrawTestfiiiPfRij")
        fn.restype = None
        fn.argtypes = [C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_uint]
        rays = C.c_int(123)
        before = tile.copy()
        fn(0.0, 0, W, H, ptr(tile), C.byref(rays), FLAG_PROGRESSIVE)
        assert len(seen) == 1 and seen[0][0] == b"DrawTest" and b"tptUpdate" in seen[0][1], seen
        assert rays.value == 0 and tile.tobytes() == before.tobytes()
    finally:
        tpt.set_error_handler(None)
    streaming(frames=3)


def row_serial_streaming(frames=4, w=24, h=12):
    """the reference's own seed mode through the device-tile path, one launch per frame (the lane-refill kernel, one lane per row)"""
    reset()
    tpt.set_seed_mode(SEED_ROW_SERIAL)
    tile = np.zeros((h, w, 4), np.float32)
    r0 = tpt.ray_counter_read()
    for f in range(frames):
        tpt.UpdateTest(0.0, f, w, h, FLAG_PROGRESSIVE)
        tpt.draw_device(0.0, f, w, h, ptr(tile), FLAG_PROGRESSIVE)
    tpt.synchronize()
    total, want, _ = oracle_frames(o, w, h, SPP, frames, seed_mode=SEED_ROW_SERIAL)
    same(tile, want, "row-serial seeds, streaming")
    assert tpt.ray_counter_read() - r0 == total


def fuzz(seed, episodes=24):
    """A seeded random walk over the API: every episode picks a frame shape, sample count, seed mode, kernel variant, pipeline depth,
    look-ahead, stream batching, a calling pattern (stream, synchronise every frame, DrawTest on a host pointer, batches) and a number of
    frames, renders them from frame 0 into a fresh buffer WITHOUT re-initialising the library in between -- whatever the previous
    episode left in flight, traced ahead, batched or allocated is the next one's starting state -- and holds image and rays against the oracle."""
    rng = np.random.default_rng(seed)
    tpt.set_scene()
    tpt.set_camera()
    s, m = o.default_scene()
    log = []
    for ep in range(episodes):
        w, h = [(24, 12), (32, 24), (40, 40), (64, 40), (72, 52)][rng.integers(0, 5)]
        spp = int(rng.integers(1, 4))
        seed_mode = SEED_ROW_SERIAL if rng.random() < 0.25 else SEED_PER_PIXEL
        pattern = ["stream", "sync", "drawtest", "batches", "sharded"][rng.integers(0, 5)]
        frames = int(rng.integers(1, 13))
        if seed_mode == SEED_ROW_SERIAL and pattern == "drawtest":
            frames = int(rng.integers(1, 40))
        persistent = 3 if rng.random() < 0.8 else 1
        if pattern == "batches" and seed_mode == SEED_PER_PIXEL:
            persistent = 3  # (per-pixel batches need the path-queue kernel: anything else is refused, by design)
        if pattern == "sharded":
            seed_mode, persistent = SEED_PER_PIXEL, 3
        if rng.random() < 0.2:  # another scene (300 spheres: the grouped traversal), or back to the built-in one
            if rng.random() < 0.5:
                from toypathtracer_amd.scenes import stress_scene
                s, m = stress_scene(300, int(rng.integers(1, 30)))
                tpt.set_scene(s, m)
            else:
                s, m = o.default_scene()
                tpt.set_scene()
        overlap = int([16, 16, 8, 2, 1][rng.integers(0, 5)])
        look = int(rng.integers(0, 4))
        sb = int(rng.random() < 0.7)
        log.append((ep, w, h, spp, seed_mode, pattern, frames, persistent, overlap, look, sb))
        tpt.set_samples_per_pixel(spp)
        tpt.set_seed_mode(seed_mode)
        tpt.set_kernel_variant(0, persistent, -1)
        tpt.set_frame_overlap(overlap)
        tpt.set_host_lookahead(look)
        tpt.set_stream_batching(sb)
        buf = np.zeros((h, w, 4), np.float32)
        cam = o.default_camera(w, h)
        want = np.zeros((h, w, 4), np.float32)
        per_frame = [o.render(s, m, cam, w, h, spp, f, FLAG_PROGRESSIVE, backbuffer=want, seed_mode=seed_mode)[0] for f in range(frames)]
        r0 = tpt.ray_counter_read()
        try:
            if pattern == "drawtest":
                for f in range(frames):
                    tpt.UpdateTest(0.0, f, w, h, FLAG_PROGRESSIVE)
                    rays = tpt.DrawTest(0.0, f, w, h, buf, FLAG_PROGRESSIVE)
                    assert rays == per_frame[f], ("rays of frame", f, rays, per_frame[f])
            elif pattern == "sharded":
                n, stripe = int(rng.integers(2, 5)), int([4, 8][rng.integers(0, 2)])
                tpt.comm_init_loopback(n, stripe)
                try:
                    f = 0
                    while f < frames:
                        k = int(min(frames - f, rng.integers(1, 4)))
                        tpt.UpdateTest(0.0, f, w, h, FLAG_PROGRESSIVE)
                        if k == 1:
                            tpt.draw_sharded(0.0, f, w, h, ptr(buf), FLAG_PROGRESSIVE)
                        else:
                            tpt.draw_sharded_batch(0.0, f, k, w, h, ptr(buf), FLAG_PROGRESSIVE)
                        f += k
                    tpt.sharded_finish()
                finally:
                    tpt.comm_destroy()
                mine = (np.arange(h) // stripe) % n == 0
                assert not buf[~mine].any(), "rows of other ranks were written"
                buf[~mine] = want[~mine]  # (a loopback communicator delivers rank 0's stripes only)
                r0 = None
            elif pattern == "batches":
                f = 0
                while f < frames:
                    k = int(min(frames - f, rng.integers(1, 6)))
                    tpt.UpdateTest(0.0, f, w, h, FLAG_PROGRESSIVE)
                    tpt.draw_device_batch(0.0, f, k, w, h, ptr(buf), FLAG_PROGRESSIVE)
                    f += k
                tpt.synchronize()
            else:
                for f in range(frames):
                    tpt.UpdateTest(0.0, f, w, h, FLAG_PROGRESSIVE)
                    tpt.draw_device(0.0, f, w, h, ptr(buf), FLAG_PROGRESSIVE)
                    if pattern == "sync":
                        tpt.synchronize()
                tpt.synchronize()
            same(buf, want, "image")
            if r0 is not None:
                total = tpt.ray_counter_read() - r0
                assert total == sum(per_frame), ("ray total", total, sum(per_frame))
        except Exception as e:  # noqa: BLE001
            try:
                tpt.synchronize()  # (nothing may stay queued on a buffer that is about to go away)
            except Exception:  # noqa: BLE001
                pass
            raise AssertionError("fuzz seed %d, episode %s: %s\n  history: %s" % (seed, log[-1], e, log[-4:])) from None
    reset()


SCENARIOS = [
    ("streaming", lambda: streaming()),
    ("streaming, one launch per frame", lambda: streaming(frames=20, batching=False)),
    ("streaming 20 frames twice, one launch per frame", lambda: (streaming(frames=20, batching=False), streaming(frames=20, w=48, h=32, batching=False))[0]),
    ("streaming 44 frames (every slot reused), one launch per frame", lambda: streaming(frames=44, w=32, h=24, batching=False)),
    ("streaming, lane-refill kernel", lambda: streaming(frames=6, hit_spheres=0, persistent=1, lds_scene=-1)),
    ("streaming, animate flag", lambda: streaming(frames=5, flags=FLAG_PROGRESSIVE | FLAG_ANIMATE)),
    ("synchronous device caller", synchronous_device_caller),
    ("DrawTest host pointer", lambda: drawtest_host()),
    ("DrawTest host pointer, no look-ahead", lambda: drawtest_host(frames=4, lookahead=0)),
    ("DrawTest reference seed mode (row-serial batches)", lambda: drawtest_host(frames=40, seed_mode=SEED_ROW_SERIAL, w=24, h=12)),
    ("look-ahead dropped by state changes", lookahead_dropped_by_state_changes),
    ("batches", lambda: batches()),
    ("batches, reference seed mode", lambda: batches((2, 5), SEED_ROW_SERIAL)),
    ("stream batching", stream_batching),
    ("animated", animated),
    ("resizes", resizes),
    ("sharded loopback n=2", lambda: sharded_loopback(2)),
    ("sharded loopback n=4", lambda: sharded_loopback(4)),
    ("sharded batches", sharded_batches),
    ("sharded frames deferred across state changes", sharded_deferred_state_changes),
    ("custom scene and camera", custom_scene_and_camera),
    ("display, caller's ray counter, tile mirror", display_counter_and_mirror),
    ("reference seed mode, streaming", row_serial_streaming),
    ("errors and re-initialisation", errors_and_reinitialisation),
]

if __name__ == "__main__":
    only = sys.argv[1:]
    for a in only:  # "fuzz:<seed>[:episodes]" adds a random walk
        if a.startswith("fuzz:"):
            parts = a.split(":")
            SCENARIOS.append((a, (lambda sd, n: (lambda: fuzz(sd, n)))(int(parts[1]), int(parts[2]) if len(parts) > 2 else 24)))
    tpt.InitializeTest()
    failed = 0
    for name, fn in SCENARIOS:
        if only and not any(k in name for k in only):
            continue
        try:
            extra = fn()
            print("OK   %s%s" % (name, "" if extra is None else "  %s" % (extra,)), flush=True)
        except Exception as e:  # noqa: BLE001
            failed += 1
            print("FAIL %s: %s: %s" % (name, type(e).__name__, e), flush=True)
    tpt.ShutdownTest()
    stats = (C.c_longlong * 2)()
    C.CDLL(os.environ["TPT_LIB"]).hostemu_stats(stats)
    h = (C.c_longlong * 3)()
    C.CDLL(os.environ["TPT_LIB"]).hostemu_helper_stats(h)
    print("operations executed %d, hipFree calls %d, policy %s; helper grids: %d found their launch closed, %d the pool dry, %d took chunks"
          % (stats[0], stats[1], os.environ.get("HOSTEMU_POLICY", "eager"), h[0], h[1], h[2]))
    sys.exit(1 if failed else 0)
