"""Is the CHECKER deterministic on this host?  Renders the same frames with oracle/tpt_oracle.c over and over (no GPU involved)
and counts results whose hash differs from the first one / from the value computed in the build container.

Round 5: two "failures" of tests/test_gpu_api.py on the GPU box turned out to have a correct GPU image and a `want` that differed
from the oracle's own output elsewhere (same binary) at 46 pixels of 24 000 (tools/mismatch_analyse.py, profiles/r05).

    python tools/oracle_determinism.py [reps] [threads ...]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from oracle_lib import SEED_PER_PIXEL, Oracle, fnv1a  # noqa: E402

EXPECT = {(200, 120, 8): None}  # filled from tests/golden if present


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    thread_sets = [int(a) for a in sys.argv[2:]] or [0]
    o = Oracle.get()
    w, h, frames = 200, 120, 8
    print("host cpus:", os.cpu_count())
    for nt in thread_sets:
        hashes = {}
        first = None
        t0 = time.time()
        for r in range(reps):
            _, bb = o.render_frames(w, h, 4, frames, seed_mode=SEED_PER_PIXEL, threads=nt)
            hsh = "%08x" % fnv1a(bb)
            if first is None:
                first = bb.copy()
            elif hsh not in hashes:
                d = (bb[..., :3] != first[..., :3]).any(axis=2)
                ys, xs = np.nonzero(d)
                print("  rep %d: new hash %s, %d pixels differ from the first result, rows %s" % (r, hsh, int(d.sum()), sorted(set(ys.tolist()))[:12]))
            hashes[hsh] = hashes.get(hsh, 0) + 1
        print("threads %d: %d reps in %.1f s -> %s" % (nt, reps, time.time() - t0, hashes))


if __name__ == "__main__":
    main()
