"""Golden vectors the REFERENCE cannot make: BASELINE.json configs[4] -- the stress scene of 4096 spheres (toypathtracer_amd/scenes.py:
the reference's scene is a static table, Test.cpp:13,46), 1920x1080, 8 spp, per-pixel seeds -- rendered by the oracle
(oracle/tpt_oracle.c, brute force over all spheres, Maths.cpp:165-202): frames 0, 1 and 2, each blended into its own zeroed tile
(what three frames in flight on three tiles give).  Merged into goldens.json as "oracle_cases"; bench.py's C5 leg and the GPU suite
compare image hashes and ray counts with them.  ~4 minutes on 8 cores.   python tests/golden/make_golden_c5.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle_lib import FLAG_PROGRESSIVE, SEED_PER_PIXEL, Oracle, fnv1a  # noqa: E402
from toypathtracer_amd.scenes import STRESS_CAMERA, stress_scene  # noqa: E402


def main():
    o = Oracle.get()
    w, h, spp = 1920, 1080, 8
    s, m = stress_scene(4096, 64)
    cam = o.camera(STRESS_CAMERA["look_from"], STRESS_CAMERA["look_at"], (0, 1, 0), STRESS_CAMERA["vfov"], w / h, STRESS_CAMERA["aperture"], STRESS_CAMERA["focus_dist"])
    out = []
    for f in range(3):
        rays, bb = o.render(s, m, cam, w, h, spp, f, FLAG_PROGRESSIVE, seed_mode=SEED_PER_PIXEL)
        out.append(dict(name="c5", scene="stress_scene(4096, 64)", width=w, height=h, spp=spp, frame=f, flags=FLAG_PROGRESSIVE, rays=int(rays), fnv="%08x" % fnv1a(bb),
                        mean_rgb=[float(bb[..., c].mean(dtype=np.float64)) for c in range(3)]))
        print(out[-1], flush=True)
    path = os.path.join(HERE, "goldens.json")
    g = json.load(open(path))
    g["oracle_cases"] = out
    g["oracle_cases_source"] = "oracle/tpt_oracle.c (per-pixel seeds, brute-force HitSpheres): the reference has no 4096-sphere scene; tests/golden/make_golden_c5.py"
    json.dump(g, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
