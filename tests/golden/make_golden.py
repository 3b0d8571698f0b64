"""Generates tests/golden/goldens.json and default_scene.npz from the PRISTINE reference build's
SCALAR path (oracle/_ref/libtpt_ref_scalar.so, made by oracle/build_ref.sh from /root/reference with
the reference's own -D__EMSCRIPTEN__ scalar switch, Config.h:9-13) -- the parity target BASELINE.json
names.  The SIMD build's hash is recorded next to it: both agree on every static-scene case and
differ on the animated one (SIMD HitSpheres breaks nearest-hit ties by SSE lane, Maths.cpp:126-159,
the scalar loop by lowest sphere id, Maths.cpp:186).  Run in the build
container (the GPU box has no /root/reference); the outputs are committed.

Harness: zeroed float buffer, for f in 0..F-1: UpdateTest(t, f, w, h, flags);
DrawTest(t, f, w, h, bb, rays, flags); hash = FNV-1a-32 over the raw bytes of all w*h*4 floats.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle_lib import FLAG_ANIMATE, FLAG_PROGRESSIVE, Ref, fnv1a  # noqa: E402

CASES = [
    # w, h, spp, frames, flags, time
    (640, 360, 4, 1, FLAG_PROGRESSIVE, 0.0),   # BASELINE.md golden a299de4a
    (640, 360, 1, 1, FLAG_PROGRESSIVE, 0.0),   # BASELINE.json config 1, golden 641c3e8f
    (1280, 720, 4, 1, FLAG_PROGRESSIVE, 0.0),  # rays 16 809 105
    (1280, 720, 4, 2, FLAG_PROGRESSIVE, 0.0),  # BASELINE.md golden 609aacda, rays 33 632 052
    (1280, 720, 4, 3, FLAG_PROGRESSIVE, 0.0),  # golden 16cce49a
    (1280, 720, 4, 10, FLAG_PROGRESSIVE, 0.0), # BASELINE.md golden 46afd557, rays 168 141 976
    (320, 180, 8, 2, FLAG_PROGRESSIVE, 0.0),
    (200, 100, 16, 2, FLAG_PROGRESSIVE, 0.0),
    (203, 117, 4, 2, FLAG_PROGRESSIVE, 0.0),   # sizes not divisible by 8
    (320, 180, 4, 2, 0, 0.0),                  # no progressive accumulation
    # kFlagAnimate moves spheres 1 and 8 in the reference's static scene (Test.cpp:306-307) and the
    # change persists in the process, so this case must stay LAST.
    (320, 180, 4, 3, FLAG_PROGRESSIVE | FLAG_ANIMATE, 0.75),
]


# The reference's other compile-time switches (Config.h:23-25), one scalar-path build each (oracle/build_ref.sh):
# variant, w, h, spp, frames, flags, time
CONFIG_CASES = [
    ("nols", 320, 180, 4, 2, FLAG_PROGRESSIVE, 0.0),                       # DO_LIGHT_SAMPLING 0
    ("mitsuba", 320, 180, 4, 2, FLAG_PROGRESSIVE, 0.0),                    # DO_MITSUBA_COMPARE 1
    ("smooth05", 320, 180, 4, 3, FLAG_PROGRESSIVE | FLAG_ANIMATE, 0.75),   # DO_ANIMATE_SMOOTHING 0.5f (moves spheres: last)
]


# The product's default seed mode (one RNG stream per pixel and frame, ComputeShader.hlsl:380 / Shaders.metal:401) made by
# REFERENCE-COMPILED code: oracle/_ref/libtpt_ref_perpixel.so is the scalar path above with that one formula put in front
# of TraceRowJob's pixel body (oracle/build_ref.sh, PERPIXEL=1).  w, h, spp, frames, flags, time
PER_PIXEL_CASES = [
    (640, 360, 1, 1, FLAG_PROGRESSIVE, 0.0),    # C1
    (640, 360, 4, 1, FLAG_PROGRESSIVE, 0.0),
    (1280, 720, 4, 1, FLAG_PROGRESSIVE, 0.0),   # C2, F = 1, 2, 3, 10
    (1280, 720, 4, 2, FLAG_PROGRESSIVE, 0.0),
    (1280, 720, 4, 3, FLAG_PROGRESSIVE, 0.0),
    (1280, 720, 4, 10, FLAG_PROGRESSIVE, 0.0),
    (1280, 720, 4, 41, FLAG_PROGRESSIVE, 0.0),  # the frames `bench.py --steps 20 --warmup 5` renders (16 priming + 5 + 20)
    (1280, 720, 4, 236, FLAG_PROGRESSIVE, 0.0), # ... and the 200-frame steady-state leg / the default `python bench.py` (16 + 20 + 200)
    (203, 117, 4, 2, FLAG_PROGRESSIVE, 0.0),    # ragged
    (203, 117, 1, 2, FLAG_PROGRESSIVE, 0.0),
    (203, 117, 8, 2, FLAG_PROGRESSIVE, 0.0),
    (203, 117, 16, 2, FLAG_PROGRESSIVE, 0.0),
    (320, 180, 4, 2, 0, 0.0),                   # no progressive accumulation
    (3840, 2160, 16, 1, FLAG_PROGRESSIVE, 0.0), # C3 / C4, whole frame
    (3840, 2160, 16, 13, FLAG_PROGRESSIVE, 0.0),# C3, the 13 frames of bench.py's secondary C3 leg (4 untimed + 9 timed)
    (320, 180, 4, 3, FLAG_PROGRESSIVE | FLAG_ANIMATE, 0.75),  # moves spheres: last
]


def per_pixel():
    ref = Ref.get("perpixel")
    out = []
    for (w, h, spp, frames, flags, t) in PER_PIXEL_CASES:
        rays, bb = ref.render_frames(w, h, spp, frames, flags, time=t)
        out.append(dict(width=w, height=h, spp=spp, frames=frames, flags=flags, time=t, rays=int(rays), fnv="%08x" % fnv1a(bb),
                        mean_rgb=[float(bb[..., c].mean(dtype=np.float64)) for c in range(3)]))
        print(out[-1])
    return out


def main():
    ref = Ref.get("scalar")
    simd = Ref.get("simd")
    ref.set_spp(4)
    ref.update(0.0, 0, 640, 360, FLAG_PROGRESSIVE)
    s, m, cam, em = ref.scene_desc()
    np.savez(os.path.join(HERE, "default_scene.npz"), spheres=s, materials=m, camera_640x360=cam, emissives=em)
    out = []
    for (w, h, spp, frames, flags, t) in CASES:
        rays, bb = ref.render_frames(w, h, spp, frames, flags, time=t)
        srays, sbb = simd.render_frames(w, h, spp, frames, flags, time=t)
        out.append(dict(width=w, height=h, spp=spp, frames=frames, flags=flags, time=t, rays=int(rays),
                        fnv="%08x" % fnv1a(bb), simd_rays=int(srays),
                        # the SIMD build's animated image is not even run-to-run stable (its ray count is)
                        simd_fnv=None if flags & FLAG_ANIMATE else "%08x" % fnv1a(sbb),
                        mean_rgb=[float(bb[..., c].mean(dtype=np.float64)) for c in range(3)],
                        alpha_max=float(bb[..., 3].max())))
        print(out[-1])
    cfg = []
    for (variant, w, h, spp, frames, flags, t) in CONFIG_CASES:
        rays, bb = Ref.get(variant).render_frames(w, h, spp, frames, flags, time=t)
        cfg.append(dict(variant=variant, width=w, height=h, spp=spp, frames=frames, flags=flags, time=t, rays=int(rays),
                        fnv="%08x" % fnv1a(bb), mean_rgb=[float(bb[..., c].mean(dtype=np.float64)) for c in range(3)]))
        print(cfg[-1])
    path = os.path.join(HERE, "goldens.json")
    keep = json.load(open(path)) if os.path.exists(path) else {}
    keep = {k: v for k, v in keep.items() if k.startswith("oracle_cases")}  # (made by make_golden_c5.py: minutes of CPU, not remade here)
    json.dump(dict(config_cases=cfg, per_pixel_cases=per_pixel(),
                   per_pixel_source="oracle/_ref/libtpt_ref_perpixel.so: the same scalar-path build with the reference's own GPU seed formula "
                                    "(ComputeShader.hlsl:380) injected at Test.cpp:281 by oracle/build_ref.sh (sed on the compiler's input stream)",
                   source="oracle/_ref/libtpt_ref_scalar.so (pristine /root/reference scalar path, g++ -O2 -ffp-contract=off "
                          "-D__EMSCRIPTEN__ -D__EMSCRIPTEN_PTHREADS__); simd_* = oracle/_ref/libtpt_ref.so",
                   cases=out, **keep), open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
