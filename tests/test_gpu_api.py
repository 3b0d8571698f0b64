"""The drop-in boundary itself: a C++ host that knows only the reference's Test API, linked against the
library; DrawTest's host-pointer contract; error behaviour."""
import os
import re
import subprocess

import numpy as np
import pytest

from oracle_lib import ROOT, FLAG_PROGRESSIVE, SEED_PER_PIXEL, SEED_ROW_SERIAL, fnv1a

pytestmark = pytest.mark.gpu


def test_cxx_host_links_and_matches_oracle(oracle, tmp_path):
    exe = str(tmp_path / "headless_host")
    libdir = os.path.join(ROOT, "toypathtracer_amd", "lib")
    subprocess.check_call(["g++", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "headless_host.cpp"),
                           "-L", libdir, "-ltoypathtracer_hip", "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.check_output([exe, "320", "180", "3"], stderr=subprocess.STDOUT).decode()
    m = re.search(r"(\d+) rays.*fnv ([0-9a-f]{8})", out)
    assert m, out
    ro, bo = oracle.render_frames(320, 180, 4, 3, seed_mode=SEED_PER_PIXEL)
    assert int(m.group(1)) == ro and m.group(2) == "%08x" % fnv1a(bo)
    assert "46 spheres, sizeof(Sphere)=20 sizeof(Material)=36 sizeof(Camera)=88" in out


def test_draw_before_update_and_bad_args_fail_cleanly(tpt_defaults):
    tpt = tpt_defaults
    with pytest.raises(tpt.TptError):
        tpt.set_samples_per_pixel(0)
    with pytest.raises(tpt.TptError):
        tpt.set_seed_mode(7)
    with pytest.raises(tpt.TptError):
        tpt.draw_device(0.0, 0, 64, 64, 0, FLAG_PROGRESSIVE)


def test_error_handler_takes_a_failure_of_the_void_api(tpt_defaults):
    """The reference's DrawTest (Test.h:14) returns void: by default a failure prints and abort()s; with tptSetErrorHandler installed the
    handler gets (entry point, message) and the call returns without effect -- through the very C++ symbol a relinked host calls."""
    import ctypes as C
    import numpy as np
    tpt = tpt_defaults
    seen = []
    tpt.set_error_handler(lambda where, msg: seen.append((where, msg)))
    try:
        lib = tpt.load_library()
        fn = getattr(lib, "_Z8DrawTestfiiiPfRij")
        fn.restype = None
        fn.argtypes = [C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_uint]
        rays = C.c_int(123)
        buf = np.full((8, 8, 4), 0.25, np.float32)
        fn(0.0, 0, 8, 8, None, C.byref(rays), FLAG_PROGRESSIVE)  # a null backbuffer: tptDraw refuses it
        assert len(seen) == 1 and seen[0][0] == b"DrawTest" and seen[0][1], seen
        assert rays.value == 0 and (buf == 0.25).all()
        tpt.UpdateTest(0.0, 0, 8, 8, FLAG_PROGRESSIVE)
        fn(0.0, 0, 8, 8, buf.ctypes.data, C.byref(rays), FLAG_PROGRESSIVE)  # ... and a good call still renders
        assert len(seen) == 1 and rays.value > 64
    finally:
        tpt.set_error_handler(None)


def test_scene_desc_round_trip(tpt_defaults, oracle):
    tpt = tpt_defaults
    tpt.UpdateTest(0.0, 0, 640, 360, FLAG_PROGRESSIVE)
    s, m, cam, em = tpt.GetSceneDesc()
    so, mo = oracle.default_scene()
    assert s.tobytes() == so.tobytes() and m.tobytes() == mo.tobytes() and list(em) == [8, 45]
    assert cam.tobytes() == oracle.default_camera(640, 360).tobytes()


def test_display_conversion_matches_reference_formula(tpt_defaults, tmp_path):
    """tptDisplayRGBA8 == Cpp/Emscripten/main.cpp:63-79 (row flip, min(sqrtf(c)*255, 255) truncated to uint8)."""
    import torch
    tpt = tpt_defaults
    w, h = 320, 180
    tile = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    for f in range(4):
        tpt.UpdateTest(0.0, f, w, h, FLAG_PROGRESSIVE)
        tpt.draw_device(0.0, f, w, h, tile.data_ptr(), FLAG_PROGRESSIVE)
    out = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    tpt.display_rgba8(tile.data_ptr(), w, h, out.data_ptr())
    tpt.synchronize()
    bb = tile.cpu().numpy()
    want = np.empty((h, w, 4), np.uint8)
    want[..., :3] = np.minimum(np.sqrt(bb[::-1, :, :3]) * np.float32(255), np.float32(255.0)).astype(np.uint8)
    want[..., 3] = 255
    got = out.cpu().numpy()
    assert np.array_equal(got, want)
    assert got[..., :3].max() > 200 and got[..., :3].min() < 80  # a real image, not a constant
    path = str(tmp_path / "frame.tga")
    tpt.write_tga(path, got)
    assert os.path.getsize(path) == 18 + w * h * 4



_LATE_HOST = r'''
import json, os, sys, time
os.environ.pop("GPU_MAX_HW_QUEUES", None)
import torch
torch.zeros(1, device="cuda").sum().item()          # the host initialises HIP FIRST: the runtime starts with its default queues
cap = sys.argv[1]
if cap != "auto":
    os.environ["TPT_OVERLAP_CAP"] = cap
sys.path.insert(0, sys.argv[2])
from toypathtracer_amd import api                    # (sets GPU_MAX_HW_QUEUES=20 -- too late for this process)
api.InitializeTest()
w, h = 1280, 720
tile = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
def burst(f0, n):
    r0 = api.ray_counter_read()
    t0 = time.perf_counter()
    for f in range(f0, f0 + n):
        api.UpdateTest(0.0, f, w, h, 2); api.draw_device(0.0, f, w, h, tile.data_ptr(), 2)
    rays = api.ray_counter_read() - r0
    return rays / (time.perf_counter() - t0) / 1e6
burst(0, 30)
rate = max(burst(30, 60), burst(90, 60))
print(json.dumps(dict(rate=rate, **api.pipeline_info())))
api.ShutdownTest()
'''


def test_host_that_initialised_hip_first_still_gets_a_working_pipeline(tmp_path):
    """GPU_MAX_HW_QUEUES is read when the HIP runtime starts; a host (or torch) that touched HIP before this library was
    loaded keeps the default of 4 hardware queues, on which 16 frames in flight run slower than 2.  tptInitialize measures
    how many streams really run side by side and clamps the frame pipeline: the automatic choice must be within 10 % of
    the best forced limit."""
    import json
    import sys
    script = tmp_path / "late_host.py"
    script.write_text(_LATE_HOST)
    res = {}
    for cap in ("auto", "1", "2", "3", "4", "8", "16"):
        out = subprocess.check_output([sys.executable, str(script), cap, ROOT], stderr=subprocess.DEVNULL, timeout=300).decode()
        res[cap] = json.loads(out.strip().splitlines()[-1])
    auto = res["auto"]
    assert auto["hw_queues"] <= 8, auto           # the probe saw the small queue pool ...
    assert auto["overlap_effective"] <= 5, auto   # ... and clamped the pipeline
    best = max(v["rate"] for v in res.values())
    assert auto["rate"] >= 0.9 * best, res


def test_pipeline_info_with_queues_set_early(tpt_defaults):
    """In the test process GPU_MAX_HW_QUEUES=20 was exported before HIP started (api.load_library): all trace streams run
    side by side, the full pipeline is used, and the per-slot buffers are allocated once per frame shape."""
    import torch
    tpt = tpt_defaults
    info = tpt.pipeline_info()
    assert info["hw_queues"] == 16 and info["overlap_effective"] == 16, info
    w, h = 320, 200
    tile = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    for f in range(0, 4):  # (a streaming caller with frames this small gets several frames per launch from its third call on:
        tpt.UpdateTest(0.0, f, w, h, FLAG_PROGRESSIVE)  #  the slots are re-reserved once for the batched colour planes)
        tpt.draw_device(0.0, f, w, h, tile.data_ptr(), FLAG_PROGRESSIVE)
    r0 = tpt.pipeline_info()["slot_reservations"]
    for f in range(4, 40):
        tpt.UpdateTest(0.0, f, w, h, FLAG_PROGRESSIVE)
        tpt.draw_device(0.0, f, w, h, tile.data_ptr(), FLAG_PROGRESSIVE)
    tpt.synchronize()
    assert tpt.pipeline_info()["slot_reservations"] == r0  # nothing allocated on the steady-state path


# ---- the host-pointer path: look-ahead, trust mode
def _draw_seq(tpt, seq, w, h, bb=None, flags=FLAG_PROGRESSIVE):
    if bb is None:
        bb = np.zeros((h, w, 4), np.float32)
    per = []
    for f in seq:
        tpt.UpdateTest(0.0, f, w, h, flags)
        per.append(tpt.DrawTest(0.0, f, w, h, bb, flags))
    return per, bb


@pytest.mark.parametrize("lookahead", [0, 1, 2, 3])
def test_drawtest_lookahead_changes_nothing(tpt_defaults, oracle, lookahead):
    """DrawTest traces the next frames ahead of its caller (tptSetHostLookahead).  Whatever the depth, every frame's bytes
    and ray count equal the oracle's -- also when the sequence does not continue as guessed: a jump in frameCount, a
    restart at 0, a different size in between."""
    from common import oracle_frames
    tpt = tpt_defaults
    tpt.set_host_lookahead(lookahead)
    w, h = 200, 120
    seq = [0, 1, 2, 3, 7, 8, 9, 0, 1, 2]           # jump 3 -> 7, restart at 0
    per, bb = _draw_seq(tpt, seq, w, h)
    s, m = oracle.default_scene()
    cam = oracle.default_camera(w, h)
    ob = np.zeros((h, w, 4), np.float32)
    want = []
    for f in seq:
        r, _ = oracle.render(s, m, cam, w, h, 4, f, seed_mode=SEED_PER_PIXEL, backbuffer=ob)
        want.append(r)
    assert per == want and bb.tobytes() == ob.tobytes()
    # another size in between, then back: the frames traced ahead for the first size are dropped
    per1, bb1 = _draw_seq(tpt, [0, 1], w, h)
    per2, bb2 = _draw_seq(tpt, [0, 1, 2], 96, 64)
    per3, bb1 = _draw_seq(tpt, [2, 3], w, h, bb1)
    ro, bo, pero = oracle_frames(oracle, w, h, 4, 4, seed_mode=SEED_PER_PIXEL)
    assert per1 + per3 == pero and bb1.tobytes() == bo.tobytes()
    ro2, bo2, pero2 = oracle_frames(oracle, 96, 64, 4, 3, seed_mode=SEED_PER_PIXEL)
    assert per2 == pero2 and bb2.tobytes() == bo2.tobytes()


def test_drawtest_in_the_reference_seed_mode_is_served_from_batched_lookahead(tpt_defaults, oracle):
    """tptSetSeedMode(0) + plain synchronous DrawTest calls -- the literal drop-in with the reference's own pixels: from the third
    consecutive frame of one configuration on the library traces that frame and the 31 after it as ONE launch (rows x frames
    lanes, a ray counter per frame), and the batch after that right behind it.  Every frame's bytes and ray count
    equal the oracle's ROW_SERIAL render, across the batch boundary (frames 33 -> 34), after a jump in frameCount, a restart
    at 0 and a different size in between."""
    from common import oracle_frames
    tpt = tpt_defaults
    tpt.set_seed_mode(SEED_ROW_SERIAL)
    tpt.set_samples_per_pixel(2)
    w, h = 160, 96
    s, m = oracle.default_scene()
    cam = oracle.default_camera(w, h)
    hits0 = tpt.lookahead_hits()
    seq = list(range(36)) + [40, 41, 0, 1]
    per, bb = _draw_seq(tpt, seq, w, h)
    ob = np.zeros((h, w, 4), np.float32)
    want = []
    for f in seq:
        r, _ = oracle.render(s, m, cam, w, h, 2, f, seed_mode=SEED_ROW_SERIAL, backbuffer=ob)
        want.append(r)
    assert per == want and bb.tobytes() == ob.tobytes()
    assert tpt.lookahead_hits() - hits0 == 33  # frames 3..35 were already traced when their call arrived (0, 1, 2 and the jumps 40, 41, 0, 1 were not: a batch is launched for a caller that has shown its pattern only)
    per2, bb2 = _draw_seq(tpt, [0, 1, 2], 96, 64)
    ro2, bo2, pero2 = oracle_frames(oracle, 96, 64, 2, 3, seed_mode=SEED_ROW_SERIAL)
    assert per2 == pero2 and bb2.tobytes() == bo2.tobytes()
    tpt.set_host_lookahead(0)  # look-ahead off: frame by frame, same bits
    per3, bb3 = _draw_seq(tpt, [0, 1, 2], 96, 64)
    assert per3 == pero2 and bb3.tobytes() == bo2.tobytes()


def test_drawtest_in_the_reference_seed_mode_survives_a_refused_batch(tpt_defaults, oracle):
    """A frame the batched launch cannot take (wider than 8192 pixels) in seed mode 0: the look-ahead is refused, DrawTest is not
    -- every call is served by the single-frame path, with the oracle's bytes and ray counts, and the refusal is remembered
    (no retry per call).  (Round 3 returned the batch's error from every DrawTest once the caller looked sequential.)"""
    from common import oracle_frames
    tpt = tpt_defaults
    tpt.set_seed_mode(SEED_ROW_SERIAL)
    tpt.set_samples_per_pixel(1)
    w, h = 8200, 2
    per, bb = _draw_seq(tpt, [0, 1, 2, 3, 4], w, h)
    ro, bo, pero = oracle_frames(oracle, w, h, 1, 5, seed_mode=SEED_ROW_SERIAL)
    assert per == pero and bb.tobytes() == bo.tobytes()


def test_stream_batching_delivers_every_frame_with_its_own_ray_count(tpt_defaults, oracle):
    """A caller that streams consecutive small frames gets several frames per launch behind its back (tptSetStreamBatching): the
    tile after EVERY frame equals the oracle's -- checked by snapshotting the tile and the ray counter with stream-ordered copies
    (no synchronise, so the caller keeps looking like a streaming one) -- across batch boundaries, a jump in frameCount and a
    change of spp in the middle; the final ray total is exact; and while only batches are in flight the running total behind
    every frame is exact too (frames served from a batch fold their rays in at their own blend, in frame order).  The same
    sequences with batching switched off give the same tiles and totals."""
    import torch
    tpt = tpt_defaults
    w, h = 256, 144  # 147 K samples at 4 spp: 8 frames per launch
    s, m = oracle.default_scene()
    cam = oracle.default_camera(w, h)

    def run(seq):
        stream = torch.cuda.Stream()
        tpt.set_stream(stream.cuda_stream)
        counter = torch.zeros(1, dtype=torch.int64, device="cuda")
        tpt.set_ray_counter(counter.data_ptr())
        tile = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
        snaps, counts = [], []
        stream.wait_stream(torch.cuda.current_stream())  # the host's own stream: ordering it against the fills is the host's job
        try:
            with torch.cuda.stream(stream):
                for f, spp in seq:
                    tpt.set_samples_per_pixel(spp)
                    tpt.UpdateTest(0.0, f, w, h, FLAG_PROGRESSIVE)
                    tpt.draw_device(0.0, f, w, h, tile.data_ptr(), FLAG_PROGRESSIVE)
                    snaps.append(tile.clone())      # stream-ordered behind this frame's blend
                    counts.append(counter.clone())
            stream.synchronize()
        finally:
            tpt.set_stream(None)
            tpt.set_ray_counter(None)
        return [t.cpu().numpy() for t in snaps], [int(c.item()) for c in counts]

    def want(seq):
        ob = np.zeros((h, w, 4), np.float32)
        tiles, totals, total = [], [], 0
        for f, spp in seq:
            r, _ = oracle.render(s, m, cam, w, h, spp, f, seed_mode=SEED_PER_PIXEL, backbuffer=ob)
            total += r
            tiles.append(ob.copy())
            totals.append(total)
        return tiles, totals

    steady = [(f, 4) for f in range(21)]                                                    # two plain calls, then batches of 8
    bumpy = [(f, 4) for f in range(5)] + [(30, 4), (31, 4), (32, 4)] + [(33, 2), (34, 2), (35, 2), (36, 2)]
    for batching in (True, False):
        tpt.set_stream_batching(batching)
        for seq in (steady, bumpy):
            tpt.set_samples_per_pixel(4)
            tiles, totals = want(seq)
            snaps, counts = run(seq)
            for k in range(len(seq)):
                assert snaps[k].tobytes() == tiles[k].tobytes(), (batching, k, seq[k])
            assert counts[-1] == totals[-1] and all(a <= b for a, b in zip(counts, counts[1:]))
            if batching and seq is steady:
                # calls 0 and 1 are plain launches (their kernels add their rays whenever they end, both before the third blend);
                # everything after them is served from batches
                assert counts[2:] == totals[2:], (counts, totals)
    tpt.set_stream_batching(False)


def test_drawtest_lookahead_is_dropped_by_every_state_change(tpt_defaults, oracle):
    """spp, scene, camera, seed mode, flags, the device path in between: each invalidates the frames traced ahead."""
    import torch
    from common import oracle_frames
    from toypathtracer_amd.scenes import stress_scene
    tpt = tpt_defaults
    w, h = 160, 96
    bb = np.zeros((h, w, 4), np.float32)
    ob = np.zeros((h, w, 4), np.float32)
    s, m = oracle.default_scene()
    s2, m2 = stress_scene(40, 8)
    cam = oracle.default_camera(w, h)
    tile = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    steps = [dict(), dict(), dict(spp=2), dict(spp=2), dict(scene=(s2, m2)), dict(scene=(s2, m2)), dict(flags=0), dict(device=True), dict(), dict(seed=0), dict()]
    cur = dict(spp=4, scene=(s, m), flags=FLAG_PROGRESSIVE, seed=SEED_PER_PIXEL)
    for f, st in enumerate(steps):
        spp, (ss, mm), flags, seed = st.get("spp", 4), st.get("scene", (s, m)), st.get("flags", FLAG_PROGRESSIVE), st.get("seed", SEED_PER_PIXEL)
        tpt.set_samples_per_pixel(spp)
        tpt.set_scene(ss, mm)
        tpt.set_seed_mode(seed)
        if st.get("device"):
            tpt.UpdateTest(0.0, f, w, h, flags)
            tpt.draw_device(0.0, f, w, h, tile.data_ptr(), flags)
            tpt.synchronize()
            continue
        tpt.UpdateTest(0.0, f, w, h, flags)
        got = tpt.DrawTest(0.0, f, w, h, bb, flags)
        want, _ = oracle.render(ss, mm, cam, w, h, spp, f, flags, seed_mode=seed, backbuffer=ob)
        assert got == want, (f, st)
        assert bb.tobytes() == ob.tobytes(), (f, st)
    tpt.set_scene(None)


def test_drawtest_trusted_buffer_mode(tpt_defaults, oracle):
    """tptSetHostBufferMode(1): the device tile is the source of truth, the buffer is uploaded once.  Same image and ray
    counts as the default mode; caller-owned alpha survives; and the default mode does pick up what the caller writes
    between frames (the reference's semantics) while the trusted mode, by contract, does not look."""
    from common import oracle_frames
    tpt = tpt_defaults
    w, h, frames = 200, 120, 6
    ro, bo, pero = oracle_frames(oracle, w, h, 4, frames, seed_mode=SEED_PER_PIXEL)
    for mode in (False, True):
        tpt.set_host_buffer_mode(mode)
        bb = np.zeros((h, w, 4), np.float32)
        bb[..., 3] = 0.25  # caller-owned alpha
        per, bb = _draw_seq(tpt, range(frames), w, h, bb)
        assert per == pero
        assert bb[..., :3].tobytes() == bo[..., :3].tobytes() and float(bb[..., 3].min()) == 0.25 == float(bb[..., 3].max())
    # default mode: a host that edits the buffer between frames sees its edit blended (prev * f + col * (1 - f))
    tpt.set_host_buffer_mode(False)
    bb = np.zeros((h, w, 4), np.float32)
    _draw_seq(tpt, [0, 1], w, h, bb)
    bb[:10, :, :3] = 5.0
    _draw_seq(tpt, [2], w, h, bb)
    assert float(bb[:10, :, :3].min()) > 3.0  # 5 * 2/3 + col / 3
    tpt.set_host_buffer_mode(False)


# ---- multi-GPU through the C ABI (RCCL inside the library)
def _build_multi_gpu_host(tmp_path):
    exe = str(tmp_path / "multi_gpu_host")
    libdir = os.path.join(ROOT, "toypathtracer_amd", "lib")
    subprocess.check_call(["g++", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "multi_gpu_host.cpp"),
                           "-L", libdir, "-ltoypathtracer_hip", "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def _run_multi_gpu_host(exe, ranks, w, h, frames, stripe, batch=1):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.check_output([exe, str(ranks), str(w), str(h), str(frames), str(stripe), str(batch)], stderr=subprocess.STDOUT, env=env,
                                  timeout=300).decode()
    m = re.search(r"(\d+) rays, .* fnv ([0-9a-f]{8})", out)
    assert m, out
    return int(m.group(1)), m.group(2)


def test_cxx_host_shards_through_the_c_abi_one_rank(oracle, tmp_path):
    """examples/multi_gpu_host.cpp with a communicator of ONE rank: librccl is loaded, ncclCommInitRank / ncclGather /
    the assemble kernel / the snapshot ring all run (a gather to oneself), image and ray count equal the oracle's."""
    exe = _build_multi_gpu_host(tmp_path)
    w, h, frames = 320, 184, 6
    rays, fnv = _run_multi_gpu_host(exe, 1, w, h, frames, 8)
    ro, bo = oracle.render_frames(w, h, 4, frames, seed_mode=SEED_PER_PIXEL)
    assert rays == ro and fnv == "%08x" % fnv1a(bo)
    rays5, fnv5 = _run_multi_gpu_host(exe, 1, 203, 117, 3, 5)   # ragged: 117 rows in stripes of 5
    ro5, bo5 = oracle.render_frames(203, 117, 4, 3, seed_mode=SEED_PER_PIXEL)
    assert rays5 == ro5 and fnv5 == "%08x" % fnv1a(bo5)
    raysb, fnvb = _run_multi_gpu_host(exe, 1, w, h, frames, 8, batch=4)   # 4 + 2 frames per launch and exchange: same image
    assert raysb == ro and fnvb == fnv


def test_cxx_host_shards_over_two_gpus(oracle, tmp_path):
    """The same host with two processes on two GPUs: RCCL over xGMI, one ncclGather per frame.  Skipped on a 1-GPU box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    exe = _build_multi_gpu_host(tmp_path)
    w, h, frames = 320, 184, 6
    rays, fnv = _run_multi_gpu_host(exe, 2, w, h, frames, 8)
    ro, bo = oracle.render_frames(w, h, 4, frames, seed_mode=SEED_PER_PIXEL)
    assert rays == ro and fnv == "%08x" % fnv1a(bo)
    assert _run_multi_gpu_host(exe, 2, w, h, frames, 8, batch=3) == (rays, fnv)


def test_bench_runs_on_two_ranks_over_rccl(tmp_path):
    """bench.py --gpus 2, both spellings: bare (bench.py starts its two ranks itself, the way the driver types its 1-GPU run) and
    under torch.distributed.run (backend nccl == RCCL: the path the driver's scaling run takes).  Skipped on a 1-GPU box."""
    import json
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "4", "--no-cpu-baseline"]
    for launcher in ([sys.executable], [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                        "--master-port", "29517"]):
        out = subprocess.check_output(launcher + tail, stderr=subprocess.DEVNULL, env=env, timeout=600).decode()
        line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
        assert line["n_gpus"] == 2 and line["value"] > 1000
        assert line["exchange"] == "cabi" and line["rccl_ranks"] == 2  # the driver's command times the C-ABI exchange
        assert line["parity_checked"] and line["parity_ok"]


def test_bench_exchanges_agree_at_world_size_one():
    """bench.py's two data paths at world size 1 -- no exchange (plain tptDrawDevice) and the library's own RCCL exchange
    (tptCommInit / tptDrawSharded / tptShardedFinish: a real one-rank RCCL communicator; there is no second implementation since
    round 6) -- give the same final image (FNV-1a of the float buffer) and the same ray total, and each equals the oracle (the
    bench's own checker leg)."""
    import json
    import sys
    lines = {}
    for ex in ("none", "cabi"):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--prime", "3",
                                       "--workload", "c1", "--exchange", ex, "--no-cpu-baseline", "--no-extras"], stderr=subprocess.DEVNULL, env=env,
                                      timeout=600).decode()
        lines[ex] = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    for ex, line in lines.items():
        assert line["exchange"] == ex and line["rccl_ranks"] == (0 if ex == "none" else 1)
        assert line["parity_checked"] and line["parity_ok"], (ex, line.get("parity_note"))
        assert line["image_fnv"] == lines["none"]["image_fnv"] == line["oracle_fnv"]
        assert line["run_rays"] == lines["none"]["run_rays"] == line["oracle_rays"]
        assert line["value"] > 100


def test_draw_sharded_in_process_single_rank(tpt_defaults, oracle):
    """tptCommInit / tptDrawSharded / tptShardedFinish from this process, communicator of one rank, size change in between."""
    import torch
    from common import oracle_frames
    tpt = tpt_defaults
    tpt.comm_init(tpt.comm_get_unique_id(), 1, 0, 8)
    try:
        for (w, h, frames) in [(256, 144, 5), (160, 90, 3)]:
            img = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
            r0 = tpt.sharded_finish()
            for f in range(frames):
                tpt.UpdateTest(0.0, f, w, h, FLAG_PROGRESSIVE)
                tpt.draw_sharded(0.0, f, w, h, img.data_ptr(), FLAG_PROGRESSIVE)
            rays = tpt.sharded_finish() - r0
            ro, bo, _ = oracle_frames(oracle, w, h, 4, frames, seed_mode=SEED_PER_PIXEL)
            assert rays == ro and img.cpu().numpy().tobytes() == bo.tobytes(), (w, h)
    finally:
        tpt.comm_destroy()


@pytest.mark.parametrize("n", [2, 4])
def test_loopback_rank0_of_n_matches_its_rows(tpt_defaults, oracle, n):
    """tptCommInitLoopback: this GPU as rank 0 of n.  Rank 0's stripes of the assembled image (and only those) carry the
    1-GPU render's bytes; the other ranks' rows stay zero; at n > 2 the pipeline runs 8 deep."""
    import numpy as np
    import torch
    from common import oracle_frames
    tpt = tpt_defaults
    w, h, frames, stripe = 200, 120, 20, 8
    tpt.comm_init_loopback(n, stripe)
    try:
        assert tpt.pipeline_info()["overlap_effective"] == (16 if n <= 2 else 8)
        img = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
        for f in range(frames):
            tpt.UpdateTest(0.0, f, w, h, FLAG_PROGRESSIVE)
            tpt.draw_sharded(0.0, f, w, h, img.data_ptr(), FLAG_PROGRESSIVE)
        tpt.sharded_finish()
        got = img.cpu().numpy()
    finally:
        tpt.comm_destroy()
    assert tpt.pipeline_info()["overlap_effective"] == 16
    _, bo, _ = oracle_frames(oracle, w, h, 4, frames, seed_mode=SEED_PER_PIXEL)
    want = np.frombuffer(bo.tobytes(), np.float32).reshape(h, w, 4)
    mine = (np.arange(h) // stripe) % n == 0
    from common import describe_image_mismatch
    assert got[mine].tobytes() == want[mine].tobytes() and not got[~mine].any(), describe_image_mismatch(got, want, mine)


def test_sharded_batches_match_per_frame_exchange(tpt_defaults, oracle):
    """tptDrawShardedBatch on a loopback communicator of 4: rank 0's stripes after batches of 3 + 1 + 4 frames equal the
    1-GPU render of 8 frames."""
    import numpy as np
    import torch
    from common import oracle_frames
    tpt = tpt_defaults
    w, h, stripe, n = 200, 120, 8, 4
    tpt.comm_init_loopback(n, stripe)
    try:
        img = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
        f = 0
        for k in [3, 1, 4]:
            tpt.UpdateTest(0.0, f, w, h, FLAG_PROGRESSIVE)
            tpt.draw_sharded_batch(0.0, f, k, w, h, img.data_ptr(), FLAG_PROGRESSIVE)
            f += k
        tpt.sharded_finish()
        got = img.cpu().numpy()
    finally:
        tpt.comm_destroy()
    _, bo, _ = oracle_frames(oracle, w, h, 4, f, seed_mode=SEED_PER_PIXEL)
    want = np.frombuffer(bo.tobytes(), np.float32).reshape(h, w, 4)
    mine = (np.arange(h) // stripe) % n == 0
    from common import describe_image_mismatch
    assert got[mine].tobytes() == want[mine].tobytes() and not got[~mine].any(), describe_image_mismatch(got, want, mine)


def test_deferred_sharded_frames_keep_the_configuration_they_were_accepted_under(tpt_defaults, oracle):
    """tptDrawSharded collects small tiles into batches of 4 (tptSetShardExchangeInterval, automatic).  A camera / scene / spp change,
    a change of the interval, a counter read and tptShardedFinish each issue what is pending first: rank 0's stripes and its ray total
    equal the oracle's, configuration by configuration (the CPU twin: tests/hostemu_driver.py)."""
    import numpy as np
    import torch
    from toypathtracer_amd.scenes import stress_scene
    tpt = tpt_defaults
    o = oracle
    w, h, stripe, n, spp = 200, 120, 8, 4, 4
    s, m = o.default_scene()
    cam = o.default_camera(w, h)
    s1, m1 = stress_scene(300, 18)
    cam1 = o.camera((0, 3, 9), (0, 0, 0), (0, 1, 0), 50.0, w / h, 0.0, 9.0)
    want = np.zeros((h, w, 4), np.float32)
    total = 0
    tpt.comm_init_loopback(n, stripe)
    try:
        img = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
        r0 = tpt.sharded_finish()
        for f in range(17):
            if f == 2:
                tpt.set_camera((0, 3, 9), (0, 0, 0), 50.0, 0.0, 9.0)
                cam = cam1
            if f == 5:
                tpt.set_scene(s1, m1)
                s, m = s1, m1
            if f == 7:
                tpt.set_samples_per_pixel(3)
                spp = 3
            if f == 10:
                tpt.set_shard_exchange_interval(3)
            if f == 14:
                assert tpt.ray_counter_read() - r0 == total
            tpt.UpdateTest(0.0, f, w, h, FLAG_PROGRESSIVE)
            tpt.draw_sharded(0.0, f, w, h, img.data_ptr(), FLAG_PROGRESSIVE)
            for y0 in range(0, h, stripe * n):
                r, _ = o.render(s, m, cam, w, h, spp, f, FLAG_PROGRESSIVE, backbuffer=want, seed_mode=SEED_PER_PIXEL, y0=y0, y1=min(y0 + stripe, h))
                total += r
        assert tpt.sharded_finish() - r0 == total
        got = img.cpu().numpy()
    finally:
        tpt.set_shard_exchange_interval(0)
        tpt.comm_destroy()
    mine = (np.arange(h) // stripe) % n == 0
    from common import describe_image_mismatch
    assert got[mine].tobytes() == want[mine].tobytes() and not got[~mine].any(), describe_image_mismatch(got, want, mine)


def test_synchronous_device_caller_gets_lookahead_and_the_same_bits(tpt_defaults, oracle):
    """tptDrawDevice + a synchronise after every frame (the reference's DrawTest contract on a device tile): from the third
    frame on the next frames are traced ahead; image, per-frame ray counts and totals equal the oracle's.  A caller that
    streams the same frames gets no look-ahead and the same bits."""
    import torch
    from common import oracle_frames
    tpt = tpt_defaults
    w, h, frames = 320, 200, 12
    ro, bo, per_frame = oracle_frames(oracle, w, h, 4, frames, seed_mode=SEED_PER_PIXEL)
    for synchronous in (True, False):
        tile = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
        hits0 = tpt.lookahead_hits()
        last = tpt.ray_counter_read()
        for f in range(frames):
            tpt.UpdateTest(0.0, f, w, h, FLAG_PROGRESSIVE)
            tpt.draw_device(0.0, f, w, h, tile.data_ptr(), FLAG_PROGRESSIVE)
            if synchronous:
                now = tpt.ray_counter_read()  # synchronises
                assert now - last == per_frame[f], f
                last = now
        total = tpt.ray_counter_read()
        if not synchronous:
            assert total - last == ro
        assert tile.cpu().numpy().tobytes() == bo.tobytes(), synchronous
        hits = tpt.lookahead_hits() - hits0
        if synchronous:  # (a streaming caller normally gets none; a host hiccup may make it look synchronous for a moment -- harmless)
            assert hits >= frames - 4, hits
    # a change of configuration in the middle drops what was traced ahead
    tile = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    for f in range(6):
        tpt.UpdateTest(0.0, f, w, h, FLAG_PROGRESSIVE)
        tpt.draw_device(0.0, f, w, h, tile.data_ptr(), FLAG_PROGRESSIVE)
        tpt.synchronize()
    tpt.set_samples_per_pixel(2)
    tile2 = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    for f in range(3):
        tpt.UpdateTest(0.0, f, w, h, FLAG_PROGRESSIVE)
        tpt.draw_device(0.0, f, w, h, tile2.data_ptr(), FLAG_PROGRESSIVE)
        tpt.synchronize()
    _, b2, _ = oracle_frames(oracle, w, h, 2, 3, seed_mode=SEED_PER_PIXEL)
    assert tile2.cpu().numpy().tobytes() == b2.tobytes()


def test_default_stream_fill_is_ordered_before_the_librarys_first_touch(tpt_defaults, oracle):
    """The ordering contract of the device path: the context's own stream is a blocking stream, so
    a tile filled on the legacy default stream (hipMemset, a torch op) is complete before the library's first blend reads it --
    even when that fill is still queued behind tens of milliseconds of other GPU work when tptDrawDevice is called -- and a
    default-stream read after the call sees the blended frame.  (Round 3 had every library stream non-blocking: the late fill
    wiped the first frames.  This test fails on that build.)"""
    import torch
    from common import oracle_frames
    tpt = tpt_defaults
    w, h, frames = 320, 200, 3
    _, bo, _ = oracle_frames(oracle, w, h, 4, frames, seed_mode=SEED_PER_PIXEL)
    a = torch.randn((4096, 4096), device="cuda")
    for rep in range(3):
        for _ in range(40):
            a = (a @ a).clamp_(-1.0, 1.0)  # ~50 ms of default-stream work in front of the fill
        tile = torch.full((h, w, 4), 123.0, dtype=torch.float32, device="cuda")
        tile.zero_()                         # the fill the library must not overtake
        for f in range(frames):
            tpt.UpdateTest(0.0, f, w, h, FLAG_PROGRESSIVE)
            tpt.draw_device(0.0, f, w, h, tile.data_ptr(), FLAG_PROGRESSIVE)
        got = tile.clone()                   # default stream: ordered behind the library's blends, no synchronise in between
        assert got.cpu().numpy().tobytes() == bo.tobytes(), rep
        tpt.synchronize()


@pytest.mark.gpu
def test_side_stream_fill_is_ordered_through_tptSetStream(tpt_defaults, oracle):
    """The other half of the ordering contract: the library orders itself against the LEGACY DEFAULT
    stream only.  A host that fills its tile on a side stream (a non-blocking stream, or a per-thread default stream) must hand
    that stream over with tptSetStream -- everything that touches the tile is then enqueued there -- or order it itself.
    With the call: the late fill lands before the first blend and the image is the oracle's.  Without it the library's blends
    overtake the fill, which then wipes them: shown here so that the documented failure mode stays a fact, not a guess."""
    import torch
    from common import oracle_frames
    tpt = tpt_defaults
    w, h, frames = 320, 200, 3
    _, bo, _ = oracle_frames(oracle, w, h, 4, frames, seed_mode=SEED_PER_PIXEL)
    side = torch.cuda.Stream()  # non-blocking: not ordered against the legacy default stream, nor against the library's blocking stream
    a = torch.randn((4096, 4096), device="cuda")
    torch.cuda.synchronize()
    results = {}
    for handed_over in (True, False):
        tile = torch.full((h, w, 4), 123.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        if handed_over:
            tpt.set_stream(side.cuda_stream)
        try:
            with torch.cuda.stream(side):
                b = a
                for _ in range(40):
                    b = (b @ b).clamp_(-1.0, 1.0)  # ~50 ms of side-stream work in front of the fill
                tile.zero_()                       # the fill the library must not overtake
            for f in range(frames):
                tpt.UpdateTest(0.0, f, w, h, FLAG_PROGRESSIVE)
                tpt.draw_device(0.0, f, w, h, tile.data_ptr(), FLAG_PROGRESSIVE)
            tpt.synchronize()
            side.synchronize()
            torch.cuda.synchronize()
            results[handed_over] = tile.cpu().numpy().tobytes() == bo.tobytes()
        finally:
            tpt.set_stream(None)
    assert results[True], "with tptSetStream the side-stream fill is ordered before the library's first blend"
    if tpt.pipeline_info()["hw_queues"] >= 16:  # (streams that have to share hardware queues are ordered by accident)
        assert not results[False], "without tptSetStream nothing orders a side-stream fill against the library (if this starts passing, the contract in INTEGRATION.md can be relaxed)"
