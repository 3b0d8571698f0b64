"""Helpers shared by the CPU and GPU test files."""
import json
import os

import numpy as np

from oracle_lib import FLAG_ANIMATE, FLAG_PROGRESSIVE, MATH_LIBM, MATH_TPT, Oracle  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))


def goldens():
    return json.load(open(os.path.join(HERE, "golden", "goldens.json")))["cases"]


def per_pixel_goldens():
    """Golden vectors of the product's default seed mode made by REFERENCE-COMPILED code: the reference's scalar CPU path with its own
    GPU seed formula (ComputeShader.hlsl:380) put in front of TraceRowJob's pixel body by oracle/build_ref.sh (PERPIXEL=1)."""
    return json.load(open(os.path.join(HERE, "golden", "goldens.json")))["per_pixel_cases"]


def oracle_goldens():
    """Golden vectors the reference cannot make (it has no 4096-sphere scene): configs[4] frames 0-2 rendered by oracle/tpt_oracle.c
    (tests/golden/make_golden_c5.py, minutes of CPU)."""
    return json.load(open(os.path.join(HERE, "golden", "goldens.json"))).get("oracle_cases", [])


def case_id(c):
    return "%dx%dx%d_f%d_fl%d" % (c["width"], c["height"], c["spp"], c["frames"], c["flags"])


def config_goldens():
    """Golden vectors of the reference's scalar path built with one of Config.h's other switches re-defined."""
    return json.load(open(os.path.join(HERE, "golden", "goldens.json")))["config_cases"]


def config_kwargs(o, case):
    """What a Config.h variant means for the oracle / the product: keyword arguments for Oracle.render + the camera."""
    w, h = case["width"], case["height"]
    v = case["variant"]
    if v == "nols":
        return dict(light_sampling=False), o.default_camera(w, h)
    if v == "mitsuba":  # aperture 0 (Test.cpp:312-313)
        return dict(mitsuba_compare=True), o.camera((0, 2, 3), (0, 0, 0), (0, 1, 0), 60.0, w / h, 0.0, 3.0)
    if v == "smooth05":
        return dict(animate_smoothing=0.5), o.default_camera(w, h)
    raise KeyError(v)


def golden_scene():
    z = np.load(os.path.join(HERE, "golden", "default_scene.npz"))
    return z["spheres"], z["materials"], z["camera_640x360"], z["emissives"]


def oracle_frames(o, w, h, spp, frames, flags=FLAG_PROGRESSIVE, time=0.0, spheres=None, mats=None, cam=None, **kw):
    """frames 0..frames-1 with the oracle on a zeroed buffer; applies kFlagAnimate like UpdateTest does."""
    if spheres is None:
        spheres, mats = o.default_scene()
    else:
        spheres = spheres.copy()
    if flags & FLAG_ANIMATE:
        o.animate(spheres, time)
    if cam is None:
        cam = o.default_camera(w, h)
    bb = np.zeros((h, w, 4), np.float32)
    total = 0
    per_frame = []
    for f in range(frames):
        r, _ = o.render(spheres, mats, cam, w, h, spp, f, flags, backbuffer=bb, **kw)
        total += r
        per_frame.append(r)
    return total, bb, per_frame


def rel_err(a, b):
    """per-channel relative error |a-b| / max(|b|, tiny) over RGB"""
    a = a[..., :3].astype(np.float64)
    b = b[..., :3].astype(np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-30)


def grazing_rays(spheres, n, seed=11):
    """Unit-direction rays that pass within a hair of a sphere's silhouette (|offset| from 1e-8 to 1e-3 radii, both
    signs): the discriminant of the targeted sphere is within rounding of zero -- the cases where a filter that is not
    conservative would drop a sphere the exact arithmetic accepts."""
    rng = np.random.default_rng(seed)
    cnt = len(spheres)
    idx = rng.integers(0, cnt, n)
    c = np.stack([spheres["cx"][idx], spheres["cy"][idx], spheres["cz"][idx]], 1).astype(np.float64)
    r = spheres["radius"][idx].astype(np.float64)
    nrm = rng.normal(size=(n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    tan = np.cross(nrm, rng.normal(size=(n, 3)))
    tan /= np.linalg.norm(tan, axis=1, keepdims=True)
    eps = 10.0 ** rng.uniform(-8, -3, n) * rng.choice([-1.0, 1.0], n)
    p = c + nrm * (r * (1.0 + eps))[:, None]          # a point just above / below the surface
    o = p - tan * rng.uniform(0.3, 6.0, n)[:, None]   # walk back along the tangent
    d = tan.astype(np.float32)
    d = (d / np.linalg.norm(d.astype(np.float64), axis=1, keepdims=True)).astype(np.float32)
    return np.concatenate([o.astype(np.float32), d], 1).astype(np.float32)


def matrix_scene(oracle, n, seed=3):
    """n (<= 64) spheres the matrix-core filter has a table for (every |a_k| in binary16 range): the first spheres of the
    built-in scene (ground r = 100 included), then copies of them shifted and shrunk at random."""
    s0, m0 = oracle.default_scene()
    rng = np.random.default_rng(seed)
    k = max(0, n - len(s0))
    s, m = s0[: min(n, len(s0))].copy(), m0[: min(n, len(s0))].copy()
    if k:
        idx = rng.integers(1, len(s0), k)
        es, em = s0[idx].copy(), m0[idx].copy()
        es["cx"] += rng.uniform(-3, 3, k).astype(np.float32)
        es["cy"] += rng.uniform(0.5, 3, k).astype(np.float32)
        es["cz"] += rng.uniform(-3, 3, k).astype(np.float32)
        es["radius"] *= rng.uniform(0.3, 1.0, k).astype(np.float32)
        em["emissive"][:] = 0  # keep the light list short
        s, m = np.concatenate([s, es]), np.concatenate([m, em])
    s["invRadius"] = np.float32(1.0) / s["radius"]
    return s, m


def describe_image_mismatch(got, want, mine=None):
    """What is wrong with a rendered image, for an assertion message: how many pixels / rows differ from the oracle's (within the
    rows `mine` when given), whether rows outside `mine` carry anything, a few sample values, and whether the differing pixels
    equal the oracle's value of another row set (a frame rendered with another sharding) or are simply zero / unblended."""
    import numpy as np
    g, w = got[..., :3], want[..., :3]
    bad = (g != w).any(axis=2)
    if mine is not None:
        inside, outside = bad & mine[:, None], got[~mine].any(axis=2)
    else:
        inside, outside = bad, np.zeros((0, 0), bool)
    rows = np.nonzero(inside.any(axis=1))[0]
    msg = ["%d pixels in %d rows differ from the oracle (rows %s...)" % (int(inside.sum()), len(rows), rows[:10].tolist())]
    ys, xs = np.nonzero(inside)
    for y, x in list(zip(ys, xs))[:3]:
        msg.append("  (x %d, y %d): got %s want %s" % (x, y, g[y, x], w[y, x]))
    if inside.any():
        msg.append("  of the differing pixels: %d are zero, %d have every channel >= the oracle's" % (int((g[inside] == 0).all(axis=1).sum()), int((g[inside] >= w[inside]).all(axis=1).sum())))
    if mine is not None and outside.any():
        oy = np.nonzero(~mine)[0][np.nonzero(outside.any(axis=1))[0]]
        sel = np.zeros_like(bad)
        sel[~mine] = outside
        msg.append("%d non-zero pixels OUTSIDE this rank's rows (rows %s...); %d of them equal the oracle's pixel there" % (
            int(outside.sum()), oy[:10].tolist(), int((g[sel] == w[sel]).all(axis=1).sum())))
    dump = os.environ.get("TPT_MISMATCH_DUMP")  # a directory (gpurun_out/...): keep the image for an analysis off the box
    if dump:
        os.makedirs(dump, exist_ok=True)
        k = len(os.listdir(dump))
        np.save(os.path.join(dump, "mismatch_%d_got.npy" % k), got)
        np.save(os.path.join(dump, "mismatch_%d_want.npy" % k), want)
        msg.append("(images saved as %s/mismatch_%d_*.npy)" % (dump, k))
    return "\n".join(msg)
