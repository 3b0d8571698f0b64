"""ctypes bindings for the CHECKERS under oracle/ (test infrastructure only).

  Oracle  -> oracle/_build/libtpt_oracle.so  (C restatement, oracle/tpt_oracle.c)
  Ref     -> oracle/_ref/libtpt_ref_scalar.so (pristine reference, its own scalar path) / libtpt_ref.so (SIMD)

Nothing under toypathtracer_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

SEED_ROW_SERIAL, SEED_PER_PIXEL = 0, 1
MATH_LIBM, MATH_TPT = 0, 1
FOLD_RECURSIVE, FOLD_FORWARD = 0, 1
FLAG_ANIMATE, FLAG_PROGRESSIVE = 1, 2

SPHERE_DT = np.dtype([("cx", "<f4"), ("cy", "<f4"), ("cz", "<f4"), ("radius", "<f4"), ("invRadius", "<f4")])
MATERIAL_DT = np.dtype([("type", "<i4"), ("albedo", "<f4", 3), ("emissive", "<f4", 3), ("roughness", "<f4"), ("ri", "<f4")])
CAMERA_DT = np.dtype([("origin", "<f4", 3), ("lowerLeftCorner", "<f4", 3), ("horizontal", "<f4", 3), ("vertical", "<f4", 3),
                      ("uu", "<f4", 3), ("vv", "<f4", 3), ("ww", "<f4", 3), ("lensRadius", "<f4")])
assert SPHERE_DT.itemsize == 20 and MATERIAL_DT.itemsize == 36 and CAMERA_DT.itemsize == 88


class Params(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("y0", C.c_int), ("y1", C.c_int), ("spp", C.c_int),
                ("frame", C.c_int), ("flags", C.c_uint), ("seed_mode", C.c_int), ("math_mode", C.c_int),
                ("fold_mode", C.c_int), ("threads", C.c_int), ("no_light_sampling", C.c_int), ("mitsuba_compare", C.c_int),
                ("has_animate_smoothing", C.c_int), ("animate_smoothing", C.c_float)]


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "_build/libtpt_oracle.so"])


def fnv1a(buf: np.ndarray) -> int:
    lib = Oracle.get().lib
    a = np.ascontiguousarray(buf)
    return lib.tpto_fnv1a(a.ctypes.data, a.nbytes)


class Oracle:
    _inst = None

    @classmethod
    def get(cls):
        if cls._inst is None:
            cls._inst = cls()
        return cls._inst

    def __init__(self):
        path = os.path.join(ORACLE_DIR, "_build", "libtpt_oracle.so")
        if not os.path.exists(path):
            build_oracle()
        self.lib = lib = C.CDLL(path)
        lib.tpto_default_scene.restype = C.c_int
        lib.tpto_default_scene.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.tpto_animate.argtypes = [C.c_void_p, C.c_float]
        lib.tpto_update_derived.argtypes = [C.c_void_p, C.c_int]
        lib.tpto_camera.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float]
        lib.tpto_default_camera.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.tpto_render.restype = C.c_int64
        lib.tpto_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(Params), C.c_void_p]
        lib.tpto_hit_spheres.restype = C.c_int
        lib.tpto_hit_spheres.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
        lib.tpto_xorshift32.restype = C.c_uint32
        lib.tpto_xorshift32.argtypes = [C.POINTER(C.c_uint32)]
        lib.tpto_random_float01.restype = C.c_float
        lib.tpto_random_float01.argtypes = [C.POINTER(C.c_uint32)]
        for f in (lib.tpto_sinf, lib.tpto_cosf, lib.tpto_pow5f):
            f.restype = C.c_float
            f.argtypes = [C.c_float]
        lib.tpto_check_sincos_vs_libm.restype = C.c_int64
        lib.tpto_check_pow5_vs_libm.restype = C.c_int64
        lib.tpto_check_pow5_vs_libm.argtypes = [C.c_uint32]
        lib.tpto_fnv1a.restype = C.c_uint32
        lib.tpto_fnv1a.argtypes = [C.c_void_p, C.c_uint64]

    def default_scene(self):
        s = np.zeros(46, SPHERE_DT)
        m = np.zeros(46, MATERIAL_DT)
        n = self.lib.tpto_default_scene(s.ctypes.data, m.ctypes.data, 46)
        assert n == 46
        return s, m

    def animate(self, spheres, time):
        self.lib.tpto_animate(spheres.ctypes.data, time)
        self.lib.tpto_update_derived(spheres.ctypes.data, len(spheres))

    def default_camera(self, w, h):
        cam = np.zeros(1, CAMERA_DT)
        self.lib.tpto_default_camera(cam.ctypes.data, w, h)
        return cam

    def camera(self, lookfrom, lookat, vup, vfov, aspect, aperture, focus):
        cam = np.zeros(1, CAMERA_DT)
        a = [np.asarray(v, np.float32) for v in (lookfrom, lookat, vup)]
        self.lib.tpto_camera(cam.ctypes.data, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data,
                             vfov, aspect, aperture, focus)
        return cam

    # Round 5: on the 256-thread hosts of the GPU boxes the checker itself was caught out -- two "parity failures" of the GPU suite had
    # a GPU image equal to this oracle's output everywhere else and a `want` that differed at 46 of 24 000 pixels (same binary, same
    # inputs; profiles/r05/README.md, tools/mismatch_analyse.py).  The code below is pure, so a checker that cannot be trusted once is
    # run twice: on big hosts (or with TPT_ORACLE_REDUNDANT=1) every frame is rendered a second time with another thread count, and
    # a disagreement is reported on stderr and settled by a third run (per-pixel majority).  TPT_ORACLE_REDUNDANT=0 switches it off.
    redundant = os.environ.get("TPT_ORACLE_REDUNDANT", "1" if (os.cpu_count() or 1) >= 64 else "0") == "1"
    disagreements = 0

    def render(self, spheres, mats, cam, w, h, spp, frame, flags=FLAG_PROGRESSIVE, seed_mode=SEED_ROW_SERIAL,
               math_mode=MATH_TPT, fold_mode=FOLD_RECURSIVE, backbuffer=None, y0=0, y1=None, threads=0,
               light_sampling=True, mitsuba_compare=False, animate_smoothing=None):
        if backbuffer is None:
            backbuffer = np.zeros((h, w, 4), np.float32)
        assert backbuffer.dtype == np.float32 and backbuffer.flags.c_contiguous and backbuffer.size == w * h * 4

        def once(bb, nt):
            p = Params(w, h, y0, h if y1 is None else y1, spp, frame, flags, seed_mode, math_mode, fold_mode, nt,
                       0 if light_sampling else 1, 1 if mitsuba_compare else 0, 0 if animate_smoothing is None else 1,
                       0.0 if animate_smoothing is None else float(animate_smoothing))
            return int(self.lib.tpto_render(spheres.ctypes.data, mats.ctypes.data, len(spheres), cam.ctypes.data, C.byref(p), bb.ctypes.data))

        if not Oracle.redundant:
            return once(backbuffer, threads), backbuffer
        before = backbuffer.copy()  # (the frame is blended into what the buffer holds, Test.cpp:293-295)
        second = before.copy()
        rays = once(backbuffer, threads)
        rays2 = once(second, 24 if threads != 24 else 16)
        if rays != rays2 or backbuffer.tobytes() != second.tobytes():
            import sys
            third = before.copy()
            rays3 = once(third, 8)
            a, b, c = (x.view(np.uint32) for x in (backbuffer, second, third))
            n_ab, n_ac, n_bc = int((a != b).sum()), int((a != c).sum()), int((b != c).sum())
            Oracle.disagreements += 1
            msg = ("oracle_lib: THE CHECKER DISAGREES WITH ITSELF on this host (frame %d, %dx%dx%d): words differing run 1/2 %d, 1/3 %d, 2/3 %d; rays %d / %d / %d -- "
                   "taking the per-word majority" % (frame, w, h, spp, n_ab, n_ac, n_bc, rays, rays2, rays3))
            print(msg, file=sys.stderr, flush=True)
            if os.environ.get("TPT_ORACLE_LOG"):  # (pytest captures stderr of passing tests: keep the evidence in a file)
                with open(os.environ["TPT_ORACLE_LOG"], "a") as fh:
                    fh.write(msg + "\n")
            none = (a != b) & (a != c) & (b != c)
            if none.any() or (rays != rays2 and rays != rays3 and rays2 != rays3):
                # three runs, three answers: there is no majority to take -- the test that asked is inconclusive, not green
                raise AssertionError(msg + " -- and %d words have THREE different values: no majority, the checker cannot be used on this host" % int(none.sum()))
            maj = np.where(a == b, a, np.where(a == c, a, b))  # (b == c where a is the odd one out)
            backbuffer.view(np.uint32)[...] = maj
            rays = rays if rays in (rays2, rays3) else rays2
        return rays, backbuffer

    def render_frames(self, w, h, spp, frames, flags=FLAG_PROGRESSIVE, **kw):
        """Default scene + camera, frames 0..frames-1 on a zeroed buffer (the golden-vector harness)."""
        s, m = self.default_scene()
        cam = self.default_camera(w, h)
        bb = np.zeros((h, w, 4), np.float32)
        total = 0
        for f in range(frames):
            r, _ = self.render(s, m, cam, w, h, spp, f, flags, backbuffer=bb, **kw)
            total += r
        return total, bb


class Ref:
    """The pristine reference build (oracle/build_ref.sh). Not re-entrant, global scene.
    variant: "scalar" (the reference's own scalar path = the parity target), "simd" (as in the repo,
    oracle flags), "fast" (SIMD, -O3 -ffast-math: how the reference ships)."""
    _inst = {}
    FILES = {"scalar": "libtpt_ref_scalar.so", "simd": "libtpt_ref.so", "fast": "libtpt_ref_fast.so",
             # scalar path with one of Config.h's other switches re-defined (oracle/build_ref.sh)
             # scalar path seeded per pixel with the reference's GPU formula (ComputeShader.hlsl:380; oracle/build_ref.sh PERPIXEL=1)
             "perpixel": "libtpt_ref_perpixel.so",
             "nols": "libtpt_ref_nols.so", "mitsuba": "libtpt_ref_mitsuba.so", "smooth05": "libtpt_ref_smooth05.so"}

    @classmethod
    def available(cls, variant="scalar"):
        return os.path.exists(cls.path(variant))

    @classmethod
    def path(cls, variant="scalar"):
        return os.path.join(ORACLE_DIR, "_ref", cls.FILES[variant])

    @classmethod
    def get(cls, variant="scalar"):
        if variant not in cls._inst:
            cls._inst[variant] = cls(variant)
        return cls._inst[variant]

    def __init__(self, variant="scalar"):
        self.variant = variant
        self.lib = lib = C.CDLL(self.path(variant))
        lib.tptref_draw.restype = C.c_int
        lib.tptref_draw.argtypes = [C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint]
        lib.tptref_update.argtypes = [C.c_float, C.c_int, C.c_int, C.c_int, C.c_uint]
        lib.tptref_set_spp.argtypes = [C.c_int]
        lib.tptref_object_count.argtypes = [C.c_void_p] * 4
        lib.tptref_scene_desc.argtypes = [C.c_void_p] * 5
        lib.tptref_init()

    def set_spp(self, spp):
        self.lib.tptref_set_spp(spp)

    def update(self, time, frame, w, h, flags):
        self.lib.tptref_update(time, frame, w, h, flags)

    def draw(self, time, frame, w, h, backbuffer, flags):
        return self.lib.tptref_draw(time, frame, w, h, backbuffer.ctypes.data, flags)

    def object_count(self):
        v = [C.c_int() for _ in range(4)]
        self.lib.tptref_object_count(*[C.byref(x) for x in v])
        return tuple(x.value for x in v)

    def scene_desc(self):
        n, so, sm, sc = self.object_count()
        assert (so, sm, sc) == (20, 36, 88)
        s = np.zeros(n, SPHERE_DT)
        m = np.zeros(n, MATERIAL_DT)
        cam = np.zeros(1, CAMERA_DT)
        em = np.zeros(n, np.int32)
        cnt = C.c_int()
        self.lib.tptref_scene_desc(s.ctypes.data, m.ctypes.data, cam.ctypes.data, em.ctypes.data, C.byref(cnt))
        return s, m, cam, em[:cnt.value].copy()

    def render_frames(self, w, h, spp, frames, flags=FLAG_PROGRESSIVE, time=0.0):
        self.set_spp(spp)
        bb = np.zeros((h, w, 4), np.float32)
        total = 0
        for f in range(frames):
            self.update(time, f, w, h, flags)
            total += self.draw(time, f, w, h, bb, flags)
        return total, bb
