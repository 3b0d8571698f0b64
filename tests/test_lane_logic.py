"""The per-lane device logic of the product (toypathtracer_amd/csrc/tpt_math.h, tpt_trace.h,
tpt_scene.h), compiled for the HOST by this test (tests/lane_emu.cpp -> tests/_build/), against the
oracle: flattened Trace/Scatter state machine, two-phase HitSpheres, scene packing, camera.  Bit-exact.
This harness is test-only; the shipped library never executes the lane logic on the CPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from common import config_goldens, config_kwargs, oracle_frames
from oracle_lib import CAMERA_DT, FLAG_PROGRESSIVE, MATERIAL_DT, SPHERE_DT, ROOT

HS = {"two_phase": 0, "simple": 1}


def emu_frames(emu, s, m, cam, w, h, spp, frames, flags, seed, hs, fold):
    bb = np.zeros((h, w, 4), np.float32)
    rays = 0
    for f in range(frames):
        rays += emu.emu_render(s.ctypes.data, m.ctypes.data, len(s), cam.ctypes.data, w, h, 0, h, spp, f, flags, seed, hs,
                               fold, bb.ctypes.data)
    return rays, bb


def test_product_default_scene_and_camera(emu, oracle):
    s, m = np.zeros(46, SPHERE_DT), np.zeros(46, MATERIAL_DT)
    assert emu.emu_default_scene(s.ctypes.data, m.ctypes.data, 46) == 46
    so, mo = oracle.default_scene()
    assert s.tobytes() == so.tobytes() and m.tobytes() == mo.tobytes()
    for (w, h) in [(640, 360), (1280, 720), (203, 117)]:
        cam = np.zeros(1, CAMERA_DT)
        emu.emu_default_camera(cam.ctypes.data, w, h)
        assert cam.tobytes() == oracle.default_camera(w, h).tobytes()


@pytest.mark.parametrize("seed", [0, 1], ids=["row_serial", "per_pixel"])
@pytest.mark.parametrize("fold", [0, 1], ids=["recursive", "forward"])
@pytest.mark.parametrize("hs", [0, 1], ids=["two_phase", "simple"])
def test_lane_state_machine_bit_exact(emu, oracle, seed, fold, hs):
    w, h, spp, frames = 160, 96, 4, 2
    s, m = oracle.default_scene()
    cam = oracle.default_camera(w, h)
    ro, bo, _ = oracle_frames(oracle, w, h, spp, frames, seed_mode=seed, fold_mode=fold)
    re, be = emu_frames(emu, s, m, cam, w, h, spp, frames, FLAG_PROGRESSIVE, seed, hs, fold)
    assert re == ro
    assert be.tobytes() == bo.tobytes()


@pytest.mark.parametrize("n", [1, 2, 7, 33, 64, 65, 130])
def test_sphere_counts_and_chunk_boundaries(emu, oracle, n):
    """odd counts (padding pair), exactly one 64-sphere chunk, chunk + 1, several chunks"""
    from toypathtracer_amd.scenes import stress_scene
    s, m = stress_scene(max(n, 6), 8)
    s, m = s[:n].copy(), m[:n].copy()
    cam = oracle.camera((0, 3, 9), (0, 0, 0), (0, 1, 0), 60.0, 80 / 48, 0.02, 9.0)
    ro, bo = oracle.render(s, m, cam, 80, 48, 2, 1, seed_mode=1)
    re, be = emu_frames(emu, s, m, cam, 80, 48, 2, 1, FLAG_PROGRESSIVE, 1, 0, 0)
    # frame index differs (oracle frame=1): redo with the same frame
    ro, bo = oracle.render(s, m, cam, 80, 48, 2, 0, seed_mode=1)
    assert re == ro and be.tobytes() == bo.tobytes()


def test_device_math_mirror_equals_oracle_math(emu, oracle):
    rng = np.random.default_rng(1)
    for r in rng.integers(0, 1 << 24, 20000):
        a = np.float32(r) / np.float32(16777216.0) * np.float32(2.0) * np.float32(3.1415926)
        assert emu.emu_sinf(a) == oracle.lib.tpto_sinf(a) and emu.emu_cosf(a) == oracle.lib.tpto_cosf(a)
    for x in np.concatenate([rng.uniform(-0.5, 1.0, 20000).astype(np.float32), np.float32([0, 1, -0.5, 1e-6, -1e-6])]):
        assert emu.emu_pow5f(x) == oracle.lib.tpto_pow5f(x)


def test_sincos_pair_all_floats(emu, oracle):
    """tsincosf -- the pair the path calls: each of glibc's two polynomials evaluated once on unsigned operands, signs applied
    to the binary32 results, swapped by the quadrant's parity -- returns the bits of tsinf / tcosf (glibc's own branch
    structure, pinned to libm on the oracle side) for EVERY float with |y| <= 120, both signs (2.2e9 arguments; the path's
    angles are 2 pi u, u = k / 2^24).  Pure IEEE binary64 / binary32 operations, no contraction: what holds here holds on the
    device (which repeats the path's 2^24 arguments in test_gpu_math.py)."""
    import ctypes as C
    emu.emu_sincos_pair_mismatches.restype = C.c_longlong
    emu.emu_sincos_pair_mismatches.argtypes = [C.c_uint, C.c_uint, C.POINTER(C.c_float)]
    first = C.c_float(0)
    hi = int(np.float32(120.0).view(np.uint32))
    bad = emu.emu_sincos_pair_mismatches(0, hi, C.byref(first))
    assert bad == 0, (bad, first.value)
    # and against the oracle's own sinf / cosf (libm's bits) on the path's call forms
    emu.emu_sincos_pair.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    sn, cs = C.c_float(), C.c_float()
    rng = np.random.default_rng(2)
    for r in list(rng.integers(0, 1 << 24, 20000)) + [0, 1, 2, (1 << 24) - 1, 1 << 21, 1 << 22, 1 << 23, 3 << 22]:
        for a in (np.float32(r) / np.float32(16777216.0) * np.float32(2.0) * np.float32(3.1415926),
                  np.float32(2.0) * np.float32(3.1415926) * (np.float32(r) / np.float32(16777216.0))):
            emu.emu_sincos_pair(a, C.byref(sn), C.byref(cs))
            assert sn.value == oracle.lib.tpto_sinf(a) and cs.value == oracle.lib.tpto_cosf(a), a


def test_two_phase_filter_is_conservative_on_grazing_rays(emu, oracle):
    """Phase 1 of the two-phase HitSpheres is a cheaper, conservative filter (FMA chains + margin); phase 2 is the
    reference's exact arithmetic.  On rays that graze a sphere within 1e-8..1e-3 radii the two-phase result must equal
    the all-exact loop bit for bit, for the built-in scene and for the 4096-sphere stress scene (64-sphere chunking)."""
    import ctypes as C
    from common import grazing_rays
    from toypathtracer_amd.scenes import stress_scene
    emu.emu_hit_spheres.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    emu.emu_hit_spheres.restype = None
    for (s, m), n in ((oracle.default_scene(), 400000), (stress_scene(4096, 64), 20000)):
        rays = grazing_rays(s, n)
        out = []
        # 0: the product's default (matrix-core filter's restatement as phase 1 for <= 64 spheres, groups for the 4096-sphere
        # scene), 1: all-exact loop, 2: two-phase flat, 3: packed VALU filter everywhere
        for hs in (0, 1, 2, 3):
            ids, ts = np.empty(n, np.int32), np.empty(n, np.float32)
            emu.emu_hit_spheres(s.ctypes.data, m.ctypes.data, len(s), hs, rays.ctypes.data, n, ids.ctypes.data, ts.ctypes.data)
            out.append((ids, ts))
        for k in (0, 2, 3):
            assert np.array_equal(out[k][0], out[1][0])
            assert np.array_equal(out[k][1].view(np.uint32), out[1][1].view(np.uint32))
        assert (out[1][0] >= 0).mean() > 0.3  # the generator does produce hits (and near misses)


@pytest.mark.parametrize("n", [256, 1000, 4096, 20000])
def test_grouped_hit_world_equals_brute_force(emu, oracle, n):
    """Large scenes are traversed through compact groups of <= 8 spheres with bounding spheres: a
    small render of the stress scene must equal the oracle's brute force bit for bit, for the grouped path (hs 0), the
    all-exact loop (hs 1) and the flat two-phase loop (hs 2)."""
    from toypathtracer_amd.scenes import STRESS_CAMERA, stress_scene
    s, m = stress_scene(n, 64 if n <= 4096 and n > 256 else (16 if n <= 256 else 160))  # 20000: several 256-group super-chunks
    w, h, spp = 64, 36, 2
    cam = oracle.camera(STRESS_CAMERA["look_from"], STRESS_CAMERA["look_at"], (0, 1, 0), STRESS_CAMERA["vfov"], w / h,
                        STRESS_CAMERA["aperture"], STRESS_CAMERA["focus_dist"])
    ro, bo = oracle.render(s, m, cam, w, h, spp, 0, seed_mode=1)
    import ctypes as C
    info = np.zeros(3, np.int32)
    emu.emu_group_info.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    emu.emu_group_info.restype = None
    emu.emu_group_info(s.ctypes.data, m.ctypes.data, n, info.ctypes.data)
    assert info[0] >= (n - info[2] + 7) // 8 and 1 <= info[2] <= 64  # grouped: ground + lights big, the rest in <= 8s
    for hs in (0, 1, 2):
        re, be = emu_frames(emu, s, m, cam, w, h, spp, 1, FLAG_PROGRESSIVE, 1, hs, 0)
        assert re == ro and be.tobytes() == bo.tobytes(), hs


def test_grouped_hit_world_breaks_ties_by_lowest_index(emu):
    """Coincident spheres give equal t: the reference keeps the lowest index (ascending loop, strict t < hitT).  The grouped
    traversal visits spheres in another order and must still return that one."""
    import ctypes as C
    from oracle_lib import MATERIAL_DT, SPHERE_DT
    emu.emu_hit_spheres.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    emu.emu_hit_spheres.restype = None
    rng = np.random.default_rng(3)
    n = 600
    s = np.zeros(n, SPHERE_DT); m = np.zeros(n, MATERIAL_DT)
    s["cx"] = rng.uniform(-20, 20, n); s["cy"] = rng.uniform(-1, 1, n); s["cz"] = rng.uniform(-20, 20, n)
    s["radius"] = rng.uniform(0.3, 0.5, n)
    dup = rng.permutation(n)[:200]                      # 100 pairs of coincident spheres, far apart in index
    for a, b in zip(dup[:100], dup[100:]):
        s[b] = s[a]
    s["invRadius"] = 1.0 / s["radius"]
    k = 20000
    tgt = rng.integers(0, n, k)
    o = np.stack([rng.uniform(-25, 25, k), rng.uniform(3, 8, k), rng.uniform(-25, 25, k)], 1).astype(np.float32)
    c = np.stack([s["cx"][tgt], s["cy"][tgt], s["cz"][tgt]], 1)
    d = c + rng.normal(0, 0.2, (k, 3)) - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    d = (d / np.linalg.norm(d.astype(np.float64), axis=1, keepdims=True)).astype(np.float32)
    rays = np.concatenate([o, d], 1).astype(np.float32)
    out = []
    for hs in (0, 1):
        ids, ts = np.empty(k, np.int32), np.empty(k, np.float32)
        emu.emu_hit_spheres(s.ctypes.data, m.ctypes.data, n, hs, rays.ctypes.data, k, ids.ctypes.data, ts.ctypes.data)
        out.append((ids, ts))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1].view(np.uint32), out[1][1].view(np.uint32))
    hit_dups = np.isin(out[1][0], dup).sum()
    assert hit_dups > 1000  # the duplicated spheres are hit often, so the tie rule is exercised


def test_filters_never_miss_in_an_adversarial_search(tmp_path):
    """tests/adversarial_filter.cpp: 60 M near-tangent ray/sphere configurations over six decades of scale (10^10 were run
    once by hand: 0 misses); neither the sphere filter nor the group-bound filter may reject what the reference accepts."""
    import subprocess
    exe = str(tmp_path / "adv")
    subprocess.check_call(["g++", "-O2", "-fopenmp", "-std=c++17", "-ffp-contract=off", "-Wno-unknown-pragmas",
                           "-I", os.path.join(ROOT, "toypathtracer_amd", "csrc"), os.path.join(ROOT, "tests", "adversarial_filter.cpp"), "-o", exe])
    out = subprocess.run([exe, "60000000"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout
    assert "FILTER MISSES 0" in out.stdout and "GROUP FILTER MISSES 0" in out.stdout and "MATRIX FILTER MISSES 0" in out.stdout


def test_matrix_filter_restatement_renders_the_oracle_image(emu, oracle):
    """Phase 1 on the matrix cores (tpt_trace.h, phase1MatrixH) is a different conservative filter: expanded around the
    coordinate origin, every factor split into two binary16 pieces, 32 slot products summed.  With its host restatement as
    phase 1 (hs 0: what the product runs for scenes with a table) the lane logic must still render the oracle's image bit
    for bit -- default scene (46 spheres: two sphere tiles, R1 = 8) and small scenes that fill one tile, one tile exactly
    (32), and both completely (64) -- and so must the packed VALU filter (hs 3)."""
    from toypathtracer_amd.scenes import stress_scene
    w, h, spp = 96, 54, 2
    s, m = oracle.default_scene()
    cam = oracle.default_camera(w, h)
    ro, bo = oracle.render(s, m, cam, w, h, spp, 0, seed_mode=1)
    for hs in (0, 3):
        re, be = emu_frames(emu, s, m, cam, w, h, spp, 1, FLAG_PROGRESSIVE, 1, hs, 0)
        assert re == ro and be.tobytes() == bo.tobytes()
    from common import matrix_scene
    for n in (3, 32, 33, 64):
        s, m = matrix_scene(oracle, n)
        assert _matrix_masks(emu, s, m, np.zeros((1, 6), np.float32))[0] >= 0  # the scene does have a table
        ro, bo = oracle.render(s, m, cam, w, h, spp, 0, seed_mode=1)
        re, be = emu_frames(emu, s, m, cam, w, h, spp, 1, FLAG_PROGRESSIVE, 1, 0, 0)
        assert re == ro and be.tobytes() == bo.tobytes(), n
    s, m = stress_scene(40, 8)  # a 1000-unit ground sphere: outside binary16 range, no table, packed VALU filter
    assert _matrix_masks(emu, s, m, np.zeros((1, 6), np.float32))[0] < 0


def _matrix_masks(emu, s, m, rays, sums=False):
    import ctypes as C
    emu.emu_matrix_masks.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    emu.emu_matrix_masks.restype = C.c_int
    n = len(rays)
    masks = np.zeros(n, np.uint64)
    S = np.zeros((n, len(s)), np.float64) if sums else None
    T = np.zeros((n, len(s)), np.float64) if sums else None
    r1 = emu.emu_matrix_masks(s.ctypes.data, m.ctypes.data, len(s), rays.ctypes.data, n, masks.ctypes.data,
                              S.ctypes.data if sums else None, T.ctypes.data if sums else None)
    return r1, masks, S, T


def test_matrix_filter_mask_layout(emu, oracle):
    """The candidate mask lists the spheres in ascending index (bit 63 - p): every sphere the exact test hits must have
    its bit set, padding bits are clear, a ray outside binary16 range keeps every sphere, and the filter filters."""
    import ctypes as C
    from common import grazing_rays
    emu.emu_hit_spheres.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    emu.emu_hit_spheres.restype = None
    s, m = oracle.default_scene()
    n = 20000
    rays = grazing_rays(s, n)
    r1, masks, S, T = _matrix_masks(emu, s, m, rays, sums=True)
    assert r1 == 8
    ids, ts = np.empty(n, np.int32), np.empty(n, np.float32)
    emu.emu_hit_spheres(s.ctypes.data, m.ctypes.data, 46, 1, rays.ctypes.data, n, ids.ctypes.data, ts.ctypes.data)
    hit = ids >= 0
    bit = (masks[hit] >> (np.uint64(63) - ids[hit].astype(np.uint64))) & np.uint64(1)
    assert bit.all()
    assert (masks & np.uint64((1 << 18) - 1)).max() == 0  # bits of spheres 46..63 never set
    counts = np.array([bin(int(x)).count("1") for x in masks[:2000]])
    assert counts.mean() < 6  # a filter, not a pass-through
    # the exact slot sum reproduces the filter's real-number value D + m: the f32 restatement's sign agrees with it
    # wherever the sum is not within the error model's bound of zero
    bits = ((masks[:, None] >> (np.uint64(63) - np.arange(46, dtype=np.uint64))[None, :]) & np.uint64(1)).astype(bool)
    clear = np.abs(S) > 64 * 2.0 ** -24 * T
    assert np.array_equal(bits[clear], (S > 0)[clear])
    far = np.float32([[300.0, 5.0, 1.0, 0.0, 1.0, 0.0], [np.nan, 0, 0, 0, 1, 0], [0, 0, 0, np.inf, 0, 0]])  # |o|^2 > 60000 / NaN / inf
    _, mfar, _, _ = _matrix_masks(emu, s, m, far)
    assert all(int(x) == ((1 << 46) - 1) << 18 for x in mfar)


@pytest.mark.parametrize("case", config_goldens(), ids=lambda c: c["variant"])
def test_config_switches_reproduce_reference_variants(emu, oracle, case):
    """DO_LIGHT_SAMPLING 0 / DO_MITSUBA_COMPARE 1 / DO_ANIMATE_SMOOTHING 0.5f as run-time switches of the lane logic
    (tptSetConfig): in the reference's own seed mode the lane logic reproduces the golden hash of the reference's scalar
    path compiled with that macro re-defined."""
    import ctypes as C
    from oracle_lib import fnv1a
    emu.emu_set_config.argtypes = [C.c_int, C.c_float, C.c_int]
    emu.emu_set_config.restype = None
    kw, cam = config_kwargs(oracle, case)
    emu.emu_set_config(1 if kw.get("light_sampling", True) else 0, kw.get("animate_smoothing", 0.9), 1 if kw.get("mitsuba_compare") else 0)
    try:
        s, m = oracle.default_scene()
        if case["flags"] & 1:
            s = s.copy()
            oracle.animate(s, case["time"])
        rays, bb = emu_frames(emu, s, m, cam, case["width"], case["height"], case["spp"], case["frames"], case["flags"], 0, 0, 0)
        assert rays == case["rays"] and "%08x" % fnv1a(bb) == case["fnv"]
    finally:
        emu.emu_set_config(1, 0.9, 0)


def test_division_by_pi_all_significands(emu):
    """tpt_math.h's tdivByPi: the 3-instruction form with the correctly rounded reciprocal of kPI as a constant equals the IEEE
    quotient a / kPI for every one of the 2^23 significands of a, across the exponent range its guard admits ([2^-100, 2^126);
    +0 maps to +0).  (The variable-divisor form, tdivSafeNum, rests on the device run over all 2^46 significand pairs:
    tools/exhaustive/exhaustive_div.hip, profiles/r04/r04_run1.log.)"""
    import ctypes as C
    emu.emu_div_pi_mismatches.restype = C.c_longlong
    emu.emu_div_pi_mismatches.argtypes = [C.c_int, C.POINTER(C.c_uint)]
    first = C.c_uint(0)
    for e in (-100, -64, -24, -3, -1, 0, 1, 2, 3, 60, 125):
        bad = emu.emu_div_pi_mismatches(e, C.byref(first))
        assert bad == 0, (e, bad, hex(first.value))


@pytest.mark.parametrize("n,grid", [(4096, 64), (1000, 32), (20000, 160)])
def test_group_bound_matrix_filter_keeps_every_group_with_an_accepted_member(emu, n, grid):
    """Grouped scenes (>= 256 spheres): the bounding spheres go through the matrix-core filter with doubled slack
    (buildGroupMatrixTable / hitSpheresGroupedDeal).  Host restatement of that filter against the reference's discriminant of
    EVERY member sphere, for rays that graze spheres within 1e-8 .. 1e-3 radii and for random rays: a member the reference accepts
    never sits in a dropped group.  (The device's MFMA accumulation differs from the restatement's within the error model both
    are conservative under; the device itself is checked by tests/test_gpu_parity.py::test_group_matrix_filter_on_the_device.)"""
    import ctypes as C
    from common import grazing_rays
    from toypathtracer_amd.scenes import stress_scene
    s, m = stress_scene(n, grid)
    rng = np.random.default_rng(11)
    k = 1500 if n <= 4096 else 300
    o = np.stack([rng.uniform(-grid / 2, grid / 2, k), rng.uniform(0.0, 8.0, k), rng.uniform(-grid / 2, grid / 2, k)], 1)
    d = rng.normal(size=(k, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays = np.concatenate([grazing_rays(s, k, seed=3), np.concatenate([o.astype(np.float32), d], 1)], 0).astype(np.float32)
    out = np.zeros(4, np.int64)
    emu.emu_group_matrix_check.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    emu.emu_group_matrix_check(s.ctypes.data, m.ctypes.data, n, rays.ctypes.data, len(rays), out.ctypes.data)
    assert out[3] > 0 and out[2] > len(rays) and out[0] == 0, out
    assert out[1] / len(rays) < 40  # (and the filter filters: a few groups per ray, not hundreds)


@pytest.mark.parametrize("n,grid", [(4096, 64), (1000, 20), (20000, 160), (3000, 0), (6000, -1)], ids=["field4096", "field1000", "field20000", "cloud3000", "cloud6000"])
def test_half_line_bounds_filter_keeps_every_group_with_an_accepted_member(emu, n, grid):
    """The three-stage dealing drops a group / super-group whose bound lies wholly behind the ray's origin (tpt_trace.h phase1PairT<true>:
    centre behind, origin outside the bound by a margin) on top of the line test.  Against the reference's whole acceptance of every
    member (positive discriminant AND a root beyond tMin), for the rays that matter: rays that graze spheres, random rays, and rays that
    LEAVE a sphere's surface into the hemisphere above it, below it and tangentially (what a bounce or a shadow ray is), also from a
    hair outside the surface.  No accepted member sits in a dropped group; the half-line form only ever drops more than the line form;
    and it does drop a good part of what the line form keeps."""
    import ctypes as C
    from common import grazing_rays
    from toypathtracer_amd.scenes import cloud_scene, stress_scene
    if grid > 0:
        s, m = stress_scene(n, grid)
    else:  # spheres spread through a volume: bounds all around the rays
        s, m = cloud_scene(n, 12.0 if grid == 0 else 15.0, 7 if grid == 0 else 11)
        grid = 24 if grid == 0 else 30
    rng = np.random.default_rng(23)
    k = 1200 if n <= 4096 else 250
    o = np.stack([rng.uniform(-grid / 2, grid / 2, k), rng.uniform(-8.0 if n in (3000, 6000) else 0.0, 8.0, k), rng.uniform(-grid / 2, grid / 2, k)], 1)
    d = rng.normal(size=(k, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    # rays leaving the surfaces of random spheres
    idx = rng.integers(1, n, 3 * k)
    c = np.stack([s["cx"][idx], s["cy"][idx], s["cz"][idx]], 1).astype(np.float64)
    r = s["radius"][idx].astype(np.float64)
    nrm = rng.normal(size=(3 * k, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    eps = np.repeat(np.array([0.0, 1e-6, 1e-3]), k)[:, None]
    po = c + nrm * (r[:, None] * (1.0 + eps))
    dd = rng.normal(size=(3 * k, 3))
    dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    leave = np.concatenate([po, dd], 1).astype(np.float32)
    leave[:, 3:] /= np.linalg.norm(leave[:, 3:], axis=1, keepdims=True)
    rays = np.concatenate([grazing_rays(s, k, seed=5), np.concatenate([o.astype(np.float32), d], 1), leave], 0).astype(np.float32)
    out = np.zeros(5, np.int64)
    emu.emu_group_half_check.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    emu.emu_group_half_check(s.ctypes.data, m.ctypes.data, n, rays.ctypes.data, len(rays), out.ctypes.data)
    assert out[3] > len(rays) // 4 and out[0] == 0 and out[4] == 0, out
    assert out[1] < 0.9 * out[2], out  # (it filters: at least a tenth of the line form's groups go)
    # ... and rays placed on the rule's own edge: origins at (1 + delta) x the radius of a group's / super-group's bound around its centre,
    # delta from -1e-3 to +1e-2 through 0 and through the rule's margin 2^-12, directions tangential and a hair to either side
    adv = np.zeros(5, np.int64)
    emu.emu_group_half_adversarial.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint, C.c_int, C.c_void_p]
    emu.emu_group_half_adversarial(s.ctypes.data, m.ctypes.data, n, 0x5bd1e995, 10 if n <= 4096 else 5, adv.ctypes.data)
    assert adv[3] > 3000 and adv[0] == 0 and adv[4] == 0, adv


def test_half_line_bounds_filter_with_loose_bounds_and_far_origins(emu):
    """The half-line rule's margin is priced with rho = (member's distance from the bound's centre) / (its radius) at the limit the
    grouping accepts, 64, and with the reference's own rounding slack, which grows with the squared distance to the ray's origin.  A scene
    made for both: 2025 spheres of radius 0.02-0.03 half a unit apart (rho 40-60: still grouped), rays that graze them from 1 ... 1000
    units away -- and the same rays reversed, every sphere behind the origin."""
    import ctypes as C
    from toypathtracer_amd.api import MATERIAL_DT, SPHERE_DT
    rng = np.random.default_rng(5)
    g = 45
    n = g * g
    s = np.zeros(n, SPHERE_DT)
    m = np.zeros(n, MATERIAL_DT)
    ix, iz = np.meshgrid(np.arange(g), np.arange(g))
    s["cx"] = ((ix.ravel() - g / 2) * 0.5 + rng.uniform(-0.05, 0.05, n)).astype(np.float32)
    s["cz"] = ((iz.ravel() - g / 2) * 0.5 + rng.uniform(-0.05, 0.05, n)).astype(np.float32)
    s["cy"] = rng.uniform(0, 0.3, n).astype(np.float32)
    s["radius"] = (0.02 * rng.uniform(1, 1.5, n)).astype(np.float32)
    s["invRadius"] = np.float32(1) / s["radius"]
    m["albedo"] = 0.5
    info = (C.c_int * 3)()
    emu.emu_group_info.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    emu.emu_group_info(s.ctypes.data, m.ctypes.data, n, C.addressof(info))
    assert info[0] > 200 and info[2] == 0, list(info)  # grouped, no group dissolved
    k = 4000
    idx = rng.integers(0, n, k)
    c = np.stack([s["cx"][idx], s["cy"][idx], s["cz"][idx]], 1).astype(np.float64)
    dirs = rng.normal(size=(k, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    perp = rng.normal(size=(k, 3))
    perp -= (perp * dirs).sum(1, keepdims=True) * dirs
    perp /= np.linalg.norm(perp, axis=1, keepdims=True)
    o = c - dirs * (10.0 ** rng.uniform(0, 3, k))[:, None] + perp * (s["radius"][idx].astype(np.float64) * rng.uniform(0.0, 1.5, k))[:, None]
    rays = np.concatenate([o, dirs], 1).astype(np.float32)
    rays[:, 3:] /= np.linalg.norm(rays[:, 3:], axis=1, keepdims=True)
    back = rays.copy()
    back[:, 3:] *= -1
    rays = np.concatenate([rays, back], 0)
    out = np.zeros(5, np.int64)
    emu.emu_group_half_check.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    emu.emu_group_half_check(s.ctypes.data, m.ctypes.data, n, rays.ctypes.data, len(rays), out.ctypes.data)
    assert out[3] > 1000 and out[0] == 0 and out[4] == 0 and out[1] < 0.7 * out[2], out
    adv = np.zeros(5, np.int64)
    emu.emu_group_half_adversarial.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint, C.c_int, C.c_void_p]
    emu.emu_group_half_adversarial(s.ctypes.data, m.ctypes.data, n, 12345, 10, adv.ctypes.data)
    assert adv[3] > 100 and adv[0] == 0 and adv[4] == 0, adv


def _queue_frames(emu, s, m, cam, w, h, spp, frames, flags, hs):
    import ctypes as C
    fn = emu.emu_render_queue_classes
    fn.restype = C.c_int64
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 4 + [C.c_uint, C.c_int, C.c_void_p]
    bb = np.zeros((h, w, 4), np.float32)
    rays = 0
    for f in range(frames):
        rays += fn(s.ctypes.data, m.ctypes.data, len(s), cam.ctypes.data, w, h, spp, f, flags, hs, bb.ctypes.data)
    return rays, bb


@pytest.mark.parametrize("hs", [0, 1, 3], ids=["matrix_filter_restatement", "simple", "valu_filter"])
def test_queue_kernel_class_code_bit_exact(emu, oracle, hs):
    """The path-queue kernel's class code -- one straight-line function per material class (qCamera, qLambertBegin, qLightRay,
    qLightShade, qLambertE, qMetal, qDielectric, qEndTerm, qFold, qStackPush; per-material 1 / ri and schlick's r0^2 from the scene
    record) -- driven on the CPU in the order the kernel applies it to a path (tests/lane_emu.cpp: emu_render_queue_classes):
    bytes and ray counts equal the oracle's, PER_PIXEL seeds, recursive fold."""
    w, h, spp, frames = 160, 96, 4, 2
    s, m = oracle.default_scene()
    cam = oracle.default_camera(w, h)
    ro, bo, _ = oracle_frames(oracle, w, h, spp, frames, seed_mode=1, fold_mode=0)
    rq, bq = _queue_frames(emu, s, m, cam, w, h, spp, frames, FLAG_PROGRESSIVE, hs)
    assert rq == ro and bq.tobytes() == bo.tobytes()


def test_queue_kernel_class_code_on_a_grouped_scene_and_with_config_switches(emu, oracle):
    """The same on a 300-sphere scene (grouped traversal, metals and dielectrics in number) and with the reference's compile-time
    switches as run-time ones (no light sampling; Mitsuba comparison mode)."""
    import ctypes as C
    from toypathtracer_amd.scenes import stress_scene
    emu.emu_set_config.argtypes = [C.c_int, C.c_float, C.c_int]
    emu.emu_set_config.restype = None
    s, m = stress_scene(300, 18)
    w, h = 96, 54
    cam = oracle.camera((0, 3, 9), (0, 0, 0), (0, 1, 0), 60.0, w / h, 0.02, 9.0)
    ro, bo = oracle.render(s, m, cam, w, h, 2, 0, seed_mode=1)
    rq, bq = _queue_frames(emu, s, m, cam, w, h, 2, 1, FLAG_PROGRESSIVE, 0)
    assert rq == ro and bq.tobytes() == bo.tobytes()
    s, m = oracle.default_scene()
    cam = oracle.default_camera(w, h)
    for ls, mitsuba in ((0, 0), (1, 1)):
        emu.emu_set_config(ls, 0.9, mitsuba)
        try:
            ro, bo = oracle.render(s, m, cam, w, h, 4, 0, seed_mode=1, light_sampling=bool(ls), mitsuba_compare=bool(mitsuba))
            rq, bq = _queue_frames(emu, s, m, cam, w, h, 4, 1, FLAG_PROGRESSIVE, 0)
            assert rq == ro and bq.tobytes() == bo.tobytes(), (ls, mitsuba)
        finally:
            emu.emu_set_config(1, 0.9, 0)
