"""Scene helpers for tests and bench: the stress scene of BASELINE.json config 5.

The reference has no such scene (static tables, Test.cpp:13,46); this is the build's own definition,
generated identically for the oracle and for the GPU (both take the arrays through their
set-scene entry points).  All arithmetic is float32 with one XorShift32 stream (Maths.cpp:5-18).
"""
import numpy as np

from .api import MATERIAL_DT, SPHERE_DT

LAMBERT, METAL, DIELECTRIC = 0, 1, 2


class _XorShift:
    def __init__(self, seed):
        self.s = np.uint32(seed)

    def rnd(self):
        x = int(self.s)
        x ^= (x << 13) & 0xFFFFFFFF
        x ^= x >> 17
        x ^= (x << 15) & 0xFFFFFFFF
        self.s = np.uint32(x)
        return np.float32(x & 0xFFFFFF) / np.float32(16777216.0)


def stress_scene(n=4096, grid=64, seed=0x9E3779B9):
    """N spheres: id 0 ground, ids 1..n-1 on a grid x grid lattice, ids 1-4 are the only lights."""
    f = np.float32
    rng = _XorShift(seed)
    s = np.zeros(n, SPHERE_DT)
    m = np.zeros(n, MATERIAL_DT)
    s[0] = (0, -1000.5, 0, 1000, 0)
    m[0] = (LAMBERT, (0.5, 0.5, 0.5), (0, 0, 0), 0, 0)
    half = grid // 2
    for i in range(1, n):
        k = i - 1
        gx, gz = (k % grid) - half, (k // grid) - half
        r = f(0.2) + f(0.25) * rng.rnd()
        cx = f(gx) + f(0.6) * rng.rnd()
        cz = f(gz) + f(0.6) * rng.rnd()
        sel = rng.rnd()
        alb = [f(0.1) + f(0.8) * rng.rnd() for _ in range(3)]
        rough = f(0.3) * rng.rnd()
        if sel < f(0.6):
            m[i] = (LAMBERT, alb, (0, 0, 0), 0, 0)
        elif sel < f(0.85):
            m[i] = (METAL, alb, (0, 0, 0), rough, 0)
        else:
            m[i] = (DIELECTRIC, alb, (0, 0, 0), 0, 1.5)
        s[i] = (cx, r - f(0.5), cz, r, 0)
    light_xz = [(-8.0, -8.0), (8.0, -8.0), (-8.0, 8.0), (8.0, 8.0)]
    for i in range(1, min(5, n)):  # the only emissive spheres (keeps light sampling at <= 4 shadow rays per hit)
        s[i] = (light_xz[i - 1][0], 6.0, light_xz[i - 1][1], 1.5, 0)
        m[i] = (LAMBERT, (0.8, 0.8, 0.8), (30, 25, 15), 0, 0)
    s["invRadius"] = f(1.0) / s["radius"]
    return s, m


STRESS_CAMERA = dict(look_from=(0.0, 6.0, 20.0), look_at=(0.0, 0.0, 0.0), vfov=60.0, aperture=0.02, focus_dist=20.0)


def cloud_scene(n=3000, extent=12.0, seed=7, lights=8):
    """N spheres spread through the cube [-extent, extent]^3 (radii 0.15 ... 0.6, three quarters Lambert, the rest metal / glass), the
    first `lights` of them emissive: a grouped scene with bounds in every direction around the rays -- nothing like the flat field of
    stress_scene (tests, tools/grouped_soak.py)."""
    rng = np.random.default_rng(seed)
    s = np.zeros(n, SPHERE_DT)
    m = np.zeros(n, MATERIAL_DT)
    s["cx"], s["cy"], s["cz"] = (rng.uniform(-extent, extent, n).astype(np.float32) for _ in range(3))
    # (a factor of 4 in radius: a group whose members lie more than 64 radii from its centre is dissolved, and a scene with more than
    #  64 loose spheres is not grouped at all)
    s["radius"] = (0.15 * 4.0 ** rng.uniform(0.0, 1.0, n)).astype(np.float32)
    kind = rng.uniform(0, 1, n)
    m["type"] = np.where(kind < 0.75, LAMBERT, np.where(kind < 0.9, METAL, DIELECTRIC)).astype(np.int32)
    m["albedo"] = rng.uniform(0.1, 0.9, (n, 3)).astype(np.float32)
    m["roughness"] = rng.uniform(0.0, 0.3, n).astype(np.float32)
    m["ri"] = np.float32(1.5)
    for i in range(min(lights, n)):
        m["type"][i] = LAMBERT
        m["emissive"][i] = (20.0, 18.0, 12.0)
        s["radius"][i] = 0.8
    s["invRadius"] = np.float32(1.0) / s["radius"]
    return s, m


CLOUD_CAMERA_INSIDE = dict(look_from=(0.5, 0.3, 0.2), look_at=(4.0, 1.0, -3.0), vfov=70.0, aperture=0.0, focus_dist=5.0)
CLOUD_CAMERA_OUTSIDE = dict(look_from=(30.0, 12.0, 28.0), look_at=(0.0, 0.0, 0.0), vfov=50.0, aperture=0.05, focus_dist=40.0)
