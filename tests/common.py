"""Helpers shared by the CPU and GPU test files."""
import json
import os

import numpy as np

from oracle_lib import FLAG_ANIMATE, FLAG_PROGRESSIVE, MATH_LIBM, MATH_TPT, Oracle  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))


def goldens():
    return json.load(open(os.path.join(HERE, "golden", "goldens.json")))["cases"]


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
