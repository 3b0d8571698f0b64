"""ctypes mirror of the reference's Test API (Cpp/Source/Test.h:10-17) over libtoypathtracer_hip.so.

Function names, argument order and meaning follow the reference so that tests read like a host of
the reference would (Cs/Program.cs:16-31, Cpp/Windows/TestWin.cpp:308-340):

    InitializeTest(); UpdateTest(t, frame, w, h, flags); rays = DrawTest(t, frame, w, h, backbuffer, flags)

Only plain pointers/ints/floats cross the boundary (include/tpt_hip.h).  Errors raise TptError;
nothing here renders on the CPU.
"""
import ctypes as C
import os

import numpy as np

kFlagAnimate = 1 << 0      # Test.h:6
kFlagProgressive = 1 << 1  # Test.h:7
SEED_ROW_SERIAL, SEED_PER_PIXEL = 0, 1
FOLD_RECURSIVE, FOLD_FORWARD = 0, 1

# layout contract of the reference (TestWin.cpp:132-134): 20 / 36 / 88 bytes
SPHERE_DT = np.dtype([("cx", "<f4"), ("cy", "<f4"), ("cz", "<f4"), ("radius", "<f4"), ("invRadius", "<f4")])
MATERIAL_DT = np.dtype([("type", "<i4"), ("albedo", "<f4", 3), ("emissive", "<f4", 3), ("roughness", "<f4"), ("ri", "<f4")])
CAMERA_DT = np.dtype([("origin", "<f4", 3), ("lowerLeftCorner", "<f4", 3), ("horizontal", "<f4", 3), ("vertical", "<f4", 3),
                      ("uu", "<f4", 3), ("vv", "<f4", 3), ("ww", "<f4", 3), ("lensRadius", "<f4")])


class TptError(RuntimeError):
    pass


_lib = None        # the library the module's functions currently talk to
_product = None    # libtoypathtracer_hip.so
_hooks = None      # libtoypathtracer_hip_hooks.so (the same sources + include/tpt_test_hooks.h), loaded by using_hooks()

# every symbol include/tpt_hip.h declares (checked by tests/test_abi.py)
C_ABI_SYMBOLS = [
    "tptInitialize", "tptShutdown", "tptUpdate", "tptDraw", "tptGetObjectCount", "tptGetSceneDesc",
    "tptSetSamplesPerPixel", "tptSetConfig", "tptSetSeedMode", "tptSetFoldMode", "tptSetScene", "tptSetCamera", "tptSetStream",
    "tptSetRowShard", "tptLocalRowCount", "tptLocalRowToGlobal", "tptDrawDevice", "tptRayCounterRead", "tptSetRayCounter", "tptSetFrameOverlap", "tptDisplayRGBA8", "tptKernelTimingBegin", "tptKernelTimingEnd",
    "tptSynchronize", "tptTimerBegin", "tptTimerEnd", "tptSetKernelVariant",
    "tptDrawDeviceBatch", "tptDrawShardedBatch", "tptGetLookaheadHits", "tptCommGetUniqueId", "tptCommInit", "tptCommInitLoopback", "tptCommInfo", "tptCommDestroy", "tptDrawSharded", "tptSetShardExchangeInterval", "tptShardedFinish", "tptGetLaunchInfo", "tptGetPipelineInfo", "tptGetSceneInfo", "tptSetHostBufferMode", "tptSetHostLookahead", "tptSetStreamBatching", "tptSetTileMirror", "tptGetLastError", "tptSetErrorHandler", "tptGetDeviceName",
]
# include/tpt_test_hooks.h: exported by the second build (libtoypathtracer_hip_hooks.so) only
HOOK_SYMBOLS = ["tptTestMath", "tptTestMathExhaustive", "tptTestHitSpheres", "tptTestMatrixFilter", "tptTestGroupFilter", "tptTestSetDealCapacities", "tptDebugStats", "tptDebugChunkOrder"]
# the reference's own C++ symbols (nm of the compiled Test.cpp), exported for link-level drop-in
CXX_ABI_SYMBOLS = [
    "_Z14InitializeTestv", "_Z12ShutdownTestv", "_Z10UpdateTestfiiij", "_Z8DrawTestfiiiPfRij",
    "_Z14GetObjectCountRiS_S_S_", "_Z12GetSceneDescPvS_S_S_Pi",
]


def _lib_dir():
    # TPT_LIB_DIR: a directory holding another build of BOTH libraries (csrc/build.sh with TPT_OUT_DIR / TPT_EXTRA_FLAGS)
    return os.environ.get("TPT_LIB_DIR") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")


def library_path():
    # TPT_LIB selects an alternative BUILD OF THE SAME HIP LIBRARY (e.g. the -DTPT_STATS profiling build)
    return os.environ.get("TPT_LIB") or os.path.join(_lib_dir(), "libtoypathtracer_hip.so")


def hooks_library_path():
    # (a profiling build selected with TPT_LIB carries the hooks itself: tools/build_variant.sh passes -DTPT_TEST_HOOKS)
    return os.environ.get("TPT_LIB") or os.path.join(_lib_dir(), "libtoypathtracer_hip_hooks.so")


def _bind(path, hooks):
    lib = C.CDLL(path)
    i, f, u, p = C.c_int, C.c_float, C.c_uint, C.c_void_p
    sigs = {
        "tptInitialize": [], "tptShutdown": [], "tptUpdate": [f, i, i, i, u],
        "tptDraw": [f, i, i, i, p, C.POINTER(i), u],
        "tptGetObjectCount": [C.POINTER(i)] * 4, "tptGetSceneDesc": [p, p, p, p, C.POINTER(i)],
        "tptSetSamplesPerPixel": [i], "tptSetConfig": [i, f, i], "tptSetSeedMode": [i], "tptSetFoldMode": [i], "tptSetScene": [p, p, i],
        "tptSetCamera": [p, p, f, f, f], "tptSetStream": [p], "tptSetRowShard": [i, i, i], "tptLocalRowCount": [i],
        "tptLocalRowToGlobal": [i], "tptDrawDevice": [f, i, i, i, p, u], "tptRayCounterRead": [C.POINTER(C.c_int64)],
        "tptSetRayCounter": [p], "tptSetTileMirror": [p, p], "tptSetFrameOverlap": [i], "tptDisplayRGBA8": [p, i, i, p], "tptKernelTimingBegin": [i],
        "tptKernelTimingEnd": [C.POINTER(f), C.POINTER(i)],
        "tptSynchronize": [], "tptTimerBegin": [], "tptTimerEnd": [C.POINTER(f)], "tptSetKernelVariant": [i, i, i],
        "tptGetLaunchInfo": [C.POINTER(i)] * 4, "tptGetPipelineInfo": [C.POINTER(i)] * 4, "tptGetSceneInfo": [C.POINTER(i)] * 3, "tptCommGetUniqueId": [p], "tptCommInit": [p, i, i, i], "tptCommInitLoopback": [i, i], "tptCommInfo": [C.POINTER(i)] * 3, "tptCommDestroy": [], "tptDrawSharded": [f, i, i, i, p, u], "tptSetShardExchangeInterval": [i], "tptDrawShardedBatch": [f, i, i, i, i, p, u], "tptDrawDeviceBatch": [f, i, i, i, i, p, u], "tptShardedFinish": [C.POINTER(C.c_int64)], "tptSetHostBufferMode": [i], "tptGetLookaheadHits": [C.POINTER(C.c_longlong)], "tptSetHostLookahead": [i], "tptSetStreamBatching": [i],
    }
    if hooks:
        sigs.update({"tptDebugStats": [p, i], "tptDebugChunkOrder": [p, p, i], "tptTestMath": [i, p, p, p, i], "tptTestMathExhaustive": [i, u, u, p, p],
                     "tptTestHitSpheres": [i, p, p, p, i], "tptTestMatrixFilter": [p, p, p, p, i], "tptTestGroupFilter": [p, i, p, p, p], "tptTestSetDealCapacities": [i, i, i]})
    for name, args in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = i
    lib.tptGetLastError.restype = C.c_char_p
    lib.tptGetDeviceName.restype = C.c_char_p
    return lib


def _preload():
    # one hardware queue per in-flight trace kernel (the runtime's default of 4 serialises deeper frame pipelining);
    # must be in the environment before the HIP runtime initialises
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
    try:  # torch ships its own HIP runtime: load it first so both sides share one libamdhip64 in the process
        import torch  # noqa: F401
    except ImportError:
        pass


def load_library():
    """dlopen the HIP library (built by __graft_entry__.build() / csrc/build.sh). Fails loudly."""
    global _lib, _product
    if _lib is not None:
        return _lib
    _preload()
    path = library_path()
    if not os.path.exists(path):
        raise TptError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(toypathtracer_amd/csrc/build.sh). There is no CPU fallback.")
    _product = _bind(path, hooks=bool(os.environ.get("TPT_LIB")) and _has_hooks(path))
    _lib = _product
    return _lib


def _has_hooks(path):
    try:
        return hasattr(C.CDLL(path), "tptTestMath")
    except OSError:
        return False


class using_hooks:
    """Context manager: inside it every function of this module talks to the HOOKS build of the library
    (libtoypathtracer_hip_hooks.so: the same sources plus the unit-test entry points of include/tpt_test_hooks.h), which is
    initialised on first use and has its own context -- scene, knobs and pipeline are separate from the product library's.
    Used by the GPU suite for math / HitSpheres / filter unit tests; every render test runs on the product library."""

    def __enter__(self):
        global _lib, _hooks
        load_library()
        if _hooks is None:
            path = hooks_library_path()
            if path == library_path():
                _hooks = _product
            else:
                if not os.path.exists(path):
                    raise TptError(f"{path} is missing (toypathtracer_amd/csrc/build.sh builds it beside the product library)")
                _hooks = _bind(path, hooks=True)
        self._prev = _lib
        _lib = _hooks
        _chk(_lib.tptInitialize(), "tptInitialize (hooks build)")
        return self

    def __exit__(self, *exc):
        # the hooks context is shut down again: its 16 trace streams would otherwise share the process's hardware queues with the
        # product library for the rest of the session (more streams than queues is a cliff: DESIGN 3.4)
        global _lib
        if _lib is not _product:
            _lib.tptShutdown()
        _lib = self._prev
        return False


def shutdown_hooks():
    global _hooks
    if _hooks is not None and _hooks is not _product:
        _hooks.tptShutdown()
    _hooks = None


def _hook(name):
    """an entry point of include/tpt_test_hooks.h: only the hooks build has it"""
    fn = getattr(load_library(), name, None)
    if fn is None:
        raise TptError(f"{name} is a unit-test hook (include/tpt_test_hooks.h): call it inside `with api.using_hooks():` -- the product library does not export it")
    return fn


def _chk(rc, where):
    if rc != 0:
        raise TptError(f"{where}: {_lib.tptGetLastError().decode()}")


# ---------------------------------------------------------------- the reference API
def InitializeTest():
    _chk(load_library().tptInitialize(), "InitializeTest")


def ShutdownTest():
    _chk(load_library().tptShutdown(), "ShutdownTest")


def UpdateTest(time, frameCount, screenWidth, screenHeight, testFlags):
    _chk(load_library().tptUpdate(time, frameCount, screenWidth, screenHeight, testFlags), "UpdateTest")


def DrawTest(time, frameCount, screenWidth, screenHeight, backbuffer, testFlags):
    """backbuffer: C-contiguous float32 numpy array of screenWidth*screenHeight*4, modified in place.
    Returns outRayCount."""
    assert isinstance(backbuffer, np.ndarray) and backbuffer.dtype == np.float32 and backbuffer.flags.c_contiguous
    assert backbuffer.size == screenWidth * screenHeight * 4
    rays = C.c_int(0)
    _chk(load_library().tptDraw(time, frameCount, screenWidth, screenHeight, backbuffer.ctypes.data, C.byref(rays),
                                testFlags), "DrawTest")
    return rays.value


def GetObjectCount():
    v = [C.c_int() for _ in range(4)]
    _chk(load_library().tptGetObjectCount(*[C.byref(x) for x in v]), "GetObjectCount")
    return tuple(x.value for x in v)


def GetSceneDesc():
    n, so, sm, sc = GetObjectCount()
    assert (so, sm, sc) == (SPHERE_DT.itemsize, MATERIAL_DT.itemsize, CAMERA_DT.itemsize)
    s, m, cam = np.zeros(n, SPHERE_DT), np.zeros(n, MATERIAL_DT), np.zeros(1, CAMERA_DT)
    em = np.zeros(n, np.int32)
    cnt = C.c_int()
    _chk(load_library().tptGetSceneDesc(s.ctypes.data, m.ctypes.data, cam.ctypes.data, em.ctypes.data, C.byref(cnt)),
         "GetSceneDesc")
    return s, m, cam, em[:cnt.value].copy()


# ---------------------------------------------------------------- run-time knobs / device path
def set_samples_per_pixel(spp):
    _chk(load_library().tptSetSamplesPerPixel(spp), "tptSetSamplesPerPixel")


def set_config(light_sampling=True, animate_smoothing=0.9, mitsuba_compare=False):
    _chk(load_library().tptSetConfig(1 if light_sampling else 0, animate_smoothing, 1 if mitsuba_compare else 0), "tptSetConfig")


def set_seed_mode(mode):
    _chk(load_library().tptSetSeedMode(mode), "tptSetSeedMode")


def set_fold_mode(mode):
    _chk(load_library().tptSetFoldMode(mode), "tptSetFoldMode")


ERROR_HANDLER = C.CFUNCTYPE(None, C.c_char_p, C.c_char_p)
_error_handler_keepalive = None


def set_error_handler(fn):
    """tptSetErrorHandler: fn(where: bytes, message: bytes) is called when one of the reference's void functions fails, instead of
    abort(); None restores the default."""
    global _error_handler_keepalive
    cb = ERROR_HANDLER(fn) if fn is not None else C.cast(None, ERROR_HANDLER)
    lib = load_library()
    lib.tptSetErrorHandler.argtypes = [ERROR_HANDLER]
    _chk(lib.tptSetErrorHandler(cb), "tptSetErrorHandler")
    _error_handler_keepalive = cb


def set_kernel_variant(hit_spheres=0, persistent=3, lds_scene=-1):
    _chk(load_library().tptSetKernelVariant(hit_spheres, persistent, lds_scene), "tptSetKernelVariant")


def set_scene(spheres=None, materials=None):
    lib = load_library()
    if spheres is None:
        _chk(lib.tptSetScene(None, None, 0), "tptSetScene")
        return
    s = np.ascontiguousarray(spheres, SPHERE_DT)
    m = np.ascontiguousarray(materials, MATERIAL_DT)
    assert len(s) == len(m)
    _chk(lib.tptSetScene(s.ctypes.data, m.ctypes.data, len(s)), "tptSetScene")


def set_camera(look_from=None, look_at=None, vfov=60.0, aperture=0.02, focus_dist=3.0):
    lib = load_library()
    if look_from is None:
        _chk(lib.tptSetCamera(None, None, 0.0, 0.0, 0.0), "tptSetCamera")
        return
    a = np.asarray(look_from, np.float32)
    b = np.asarray(look_at, np.float32)
    _chk(lib.tptSetCamera(a.ctypes.data, b.ctypes.data, vfov, aperture, focus_dist), "tptSetCamera")


def set_stream(stream_handle):
    _chk(load_library().tptSetStream(C.c_void_p(stream_handle) if stream_handle else None), "tptSetStream")


def set_row_shard(stripe_rows, num_parts, part):
    _chk(load_library().tptSetRowShard(stripe_rows, num_parts, part), "tptSetRowShard")


def local_row_count(height):
    return load_library().tptLocalRowCount(height)


def local_row_to_global(local_row):
    return load_library().tptLocalRowToGlobal(local_row)


def draw_device(time, frameCount, screenWidth, screenHeight, device_ptr, testFlags):
    """Asynchronous DrawTest into a device-resident tile (raw device pointer, e.g. tensor.data_ptr())."""
    _chk(load_library().tptDrawDevice(time, frameCount, screenWidth, screenHeight, C.c_void_p(device_ptr), testFlags),
         "tptDrawDevice")


def ray_counter_read():
    v = C.c_int64()
    _chk(load_library().tptRayCounterRead(C.byref(v)), "tptRayCounterRead")
    return v.value


def set_frame_overlap(frames):
    _chk(load_library().tptSetFrameOverlap(frames), "tptSetFrameOverlap")


def display_rgba8(device_tile_ptr, width, height, device_rgba_ptr):
    """linear float4 image (device) -> RGBA8 (device), the reference's sqrt display transform, top row first"""
    _chk(load_library().tptDisplayRGBA8(C.c_void_p(device_tile_ptr), width, height, C.c_void_p(device_rgba_ptr)), "tptDisplayRGBA8")


def write_tga(path, rgba):
    """rgba: uint8 [h, w, 4], top row first.  Uncompressed 32-bit TGA like the reference's only image writer
    (Cs/Program.cs:33-59: BGRA, bottom-left origin)."""
    rgba = np.ascontiguousarray(rgba, np.uint8)
    h, w = rgba.shape[:2]
    header = bytes([0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, w & 0xFF, (w >> 8) & 0xFF, h & 0xFF, (h >> 8) & 0xFF, 32, 0])
    with open(path, "wb") as f:
        f.write(header)
        f.write(rgba[::-1, :, [2, 1, 0, 3]].tobytes())


def set_ray_counter(device_ptr):
    """device_ptr: address of one zeroed int64 in device memory (tensor.data_ptr()), or None/0 for the internal one."""
    _chk(load_library().tptSetRayCounter(C.c_void_p(device_ptr) if device_ptr else None), "tptSetRayCounter")


def set_tile_mirror(mirror_ptr, counter_out_ptr=None):
    """The resolve kernel also writes the blended tile to `mirror_ptr` and the ray counter to `counter_out_ptr` (device
    addresses; None/0 turns it off)."""
    _chk(load_library().tptSetTileMirror(C.c_void_p(mirror_ptr) if mirror_ptr else None,
                                         C.c_void_p(counter_out_ptr) if counter_out_ptr else None), "tptSetTileMirror")


def synchronize():
    _chk(load_library().tptSynchronize(), "tptSynchronize")


def timer_begin():
    _chk(load_library().tptTimerBegin(), "tptTimerBegin")


def timer_end():
    ms = C.c_float()
    _chk(load_library().tptTimerEnd(C.byref(ms)), "tptTimerEnd")
    return ms.value


def kernel_timing_begin(max_launches):
    _chk(load_library().tptKernelTimingBegin(max_launches), "tptKernelTimingBegin")


def kernel_timing_end():
    """-> (sum of the individual trace-launch durations in ms, number of launches)"""
    ms, n = C.c_float(), C.c_int()
    _chk(load_library().tptKernelTimingEnd(C.byref(ms), C.byref(n)), "tptKernelTimingEnd")
    return ms.value, n.value


def launch_info():
    v = [C.c_int() for _ in range(4)]
    load_library().tptGetLaunchInfo(*[C.byref(x) for x in v])
    return dict(blocks_per_cu=v[0].value, lds_bytes=v[1].value, grid_blocks=v[2].value, num_cus=v[3].value)


def comm_get_unique_id():
    buf = (C.c_char * 128)()
    _chk(load_library().tptCommGetUniqueId(buf), "tptCommGetUniqueId")
    return bytes(buf)


def comm_init(unique_id, n_ranks, rank, stripe_rows=8):
    _chk(load_library().tptCommInit(C.c_char_p(unique_id), n_ranks, rank, stripe_rows), "tptCommInit")


def comm_init_loopback(n_ranks, stripe_rows=8):
    """Measurement aid: play rank 0 of an n_ranks-way sharded run alone (device copy in place of the gather)."""
    _chk(load_library().tptCommInitLoopback(n_ranks, stripe_rows), "tptCommInitLoopback")


def comm_info():
    """(ranks, rank, loopback) as the communicator itself reports them (ncclCommCount / ncclCommUserRank)"""
    v = [C.c_int() for _ in range(3)]
    _chk(load_library().tptCommInfo(*[C.byref(x) for x in v]), "tptCommInfo")
    return v[0].value, v[1].value, bool(v[2].value)


def comm_destroy():
    _chk(load_library().tptCommDestroy(), "tptCommDestroy")


def draw_sharded(time, frameCount, screenWidth, screenHeight, device_image_ptr, testFlags):
    _chk(load_library().tptDrawSharded(time, frameCount, screenWidth, screenHeight, device_image_ptr, testFlags), "tptDrawSharded")


def set_shard_exchange_interval(k):
    """0 = automatic (every frame for big tiles, every 2nd / 4th for small ones), k >= 1 = every k-th frame; same on every rank"""
    _chk(load_library().tptSetShardExchangeInterval(k), "tptSetShardExchangeInterval")


def draw_sharded_batch(time, firstFrame, nFrames, screenWidth, screenHeight, device_image_ptr, testFlags):
    _chk(load_library().tptDrawShardedBatch(time, firstFrame, nFrames, screenWidth, screenHeight, device_image_ptr, testFlags), "tptDrawShardedBatch")


def draw_device_batch(time, firstFrame, nFrames, screenWidth, screenHeight, device_tile_ptr, testFlags):
    """nFrames consecutive frames in one launch (static scene); same bits as nFrames draw_device calls."""
    _chk(load_library().tptDrawDeviceBatch(time, firstFrame, nFrames, screenWidth, screenHeight, device_tile_ptr, testFlags), "tptDrawDeviceBatch")


def sharded_finish():
    v = C.c_int64()
    _chk(load_library().tptShardedFinish(C.byref(v)), "tptShardedFinish")
    return v.value


def set_host_buffer_mode(only_written_by_drawtest):
    _chk(load_library().tptSetHostBufferMode(1 if only_written_by_drawtest else 0), "tptSetHostBufferMode")


def set_host_lookahead(frames):
    _chk(load_library().tptSetHostLookahead(frames), "tptSetHostLookahead")


def set_stream_batching(enable):
    _chk(load_library().tptSetStreamBatching(1 if enable else 0), "tptSetStreamBatching")


def lookahead_hits():
    v = C.c_longlong()
    _chk(load_library().tptGetLookaheadHits(C.byref(v)), "tptGetLookaheadHits")
    return v.value


def pipeline_info():
    v = [C.c_int() for _ in range(4)]
    _chk(load_library().tptGetPipelineInfo(*[C.byref(x) for x in v]), "tptGetPipelineInfo")
    return dict(hw_queues=v[0].value, overlap_effective=v[1].value, stream_depth=v[2].value, slot_reservations=v[3].value)


def scene_info():
    """what the next launch does with the scene: spheres, groups (0: flat), groups' bounds on the matrix cores (False: packed VALU filter)"""
    v = [C.c_int() for _ in range(3)]
    _chk(load_library().tptGetSceneInfo(*[C.byref(x) for x in v]), "tptGetSceneInfo")
    return dict(spheres=v[0].value, groups=v[1].value, bounds_on_matrix_cores=bool(v[2].value))


def debug_stats(reset=True):
    """profiling build only (TPT_LIB=... built with -DTPT_STATS)"""
    out = np.zeros(128, np.uint64)
    _chk(_hook("tptDebugStats")(out.ctypes.data, 1 if reset else 0), "tptDebugStats")
    return out


def debug_chunk_order(capacity=1 << 20):
    cost = np.zeros(capacity, np.uint32)
    order = np.zeros(capacity, np.uint32)
    n = _hook("tptDebugChunkOrder")(cost.ctypes.data, order.ctypes.data, capacity)
    if n < 0:
        raise TptError("tptDebugChunkOrder: " + _lib.tptGetLastError().decode())
    return cost[:n].copy(), order[:n].copy()


def device_name():
    return load_library().tptGetDeviceName().decode()


def test_math(op, a, b=None):
    a = np.ascontiguousarray(a, np.float32)
    out = np.empty_like(a)
    bp = None
    if b is not None:
        b = np.ascontiguousarray(b, np.float32)
        bp = b.ctypes.data
    _chk(_hook("tptTestMath")(op, a.ctypes.data, bp, out.ctypes.data, a.size), "tptTestMath")
    return out


def test_math_exhaustive(op, lo=0, hi=0xFFFFFFFF):
    """fast correctly-rounded sqrt (op 0) / 1/sqrt-then-reciprocal (op 1) vs the compiler's expansions for every bit
    pattern in [lo, hi], on the device -> (mismatches, first offending inputs)"""
    bad = C.c_ulonglong(0)
    first = (C.c_uint * 8)()
    _chk(_hook("tptTestMathExhaustive")(op, lo, hi, C.byref(bad), first), "tptTestMathExhaustive")
    return int(bad.value), [int(v) for v in first][: min(8, int(bad.value))]


def test_hit_spheres(rays, hit_spheres=0):
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    n = rays.shape[0]
    ids = np.empty(n, np.int32)
    ts = np.empty(n, np.float32)
    _chk(_hook("tptTestHitSpheres")(hit_spheres, rays.ctypes.data, ids.ctypes.data, ts.ctypes.data, n),
         "tptTestHitSpheres")
    return ids, ts


def test_matrix_filter(rays, hits=False):
    """phase 1 of HitSpheres on the matrix cores for the current scene (<= 64 spheres): candidate masks (sphere p at bit
    63 - p); hits=True: (masks, ids, ts) with the nearest hit through the filter + the exact test of its candidates"""
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    n = rays.shape[0]
    masks = np.empty(n, np.uint64)
    ids = np.empty(n, np.int32) if hits else None
    ts = np.empty(n, np.float32) if hits else None
    _chk(_hook("tptTestMatrixFilter")(rays.ctypes.data, masks.ctypes.data, ids.ctypes.data if hits else None,
                                            ts.ctypes.data if hits else None, n), "tptTestMatrixFilter")
    return (masks, ids, ts) if hits else masks


def test_set_deal_capacities(super_group_entries=0, group_entries=0, survivor_entries=0):
    """hooks build: run-time sizes of the three entry areas of the grouped traversal (0, 0, 0: the compiled ones)"""
    _chk(_hook("tptTestSetDealCapacities")(super_group_entries, group_entries, survivor_entries), "tptTestSetDealCapacities")


def test_group_filter(rays):
    """the matrix-core filter over the group bounds of the current grouped scene vs the reference's discriminant of every member:
    (violations, groups kept per ray, members with a positive discriminant per ray)"""
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    n = rays.shape[0]
    v = [C.c_ulonglong(0) for _ in range(3)]
    _chk(_hook("tptTestGroupFilter")(rays.ctypes.data, n, C.byref(v[0]), C.byref(v[1]), C.byref(v[2])), "tptTestGroupFilter")
    return int(v[0].value), v[1].value / n, v[2].value / n
