"""Off the GPU box: what happened to the pixels of a failed test_sharded_batches_match_per_frame_exchange /
test_loopback_rank0_of_n_matches_its_rows run (images kept by tests/common.py describe_image_mismatch under TPT_MISMATCH_DUMP)?

For every differing pixel: which single frame's colour would have to be different, and is that other colour one the oracle knows --
another frame's colour at that pixel (a stale colour plane), the frame's colour with one sample missing / counted twice, zero?

    python tools/mismatch_analyse.py gpurun_out/r05_5/dump/mismatch_0 [frames]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from oracle_lib import SEED_PER_PIXEL, Oracle  # noqa: E402


def main():
    stem = sys.argv[1]
    got, want = np.load(stem + "_got.npy"), np.load(stem + "_want.npy")
    h, w = got.shape[:2]
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    o = Oracle.get()
    s, m = o.default_scene()
    cam = o.default_camera(w, h)
    cols = []
    for f in range(frames):
        _, bb = o.render(s, m, cam, w, h, 4, f, 0, seed_mode=SEED_PER_PIXEL)
        cols.append(bb[..., :3].copy())
    # partial sums: the frame's colour if only the first k samples had been added (spp = k renders share the RNG stream of the pixel)
    part = {}
    for f in range(frames):
        for k in (1, 2, 3):
            _, bb = o.render(s, m, cam, w, h, k, f, 0, seed_mode=SEED_PER_PIXEL)
            part[(f, k)] = bb[..., :3] * np.float32(k)  # sum of the first k samples (the oracle divides by spp)

    def blend(pix_cols):
        acc = np.zeros(3, np.float32)
        for f, c in enumerate(pix_cols):
            lerp = np.float32(np.float32(f) / np.float32(f + 1))
            acc = acc * lerp + c * np.float32(np.float32(1) - lerp)
        return acc

    bad = (got[..., :3] != want[..., :3]).any(axis=2) & got[..., :3].any(axis=2)  # (rows of other ranks stay zero in a loopback run)
    ys, xs = np.nonzero(bad)
    print("%d differing pixels" % len(ys))
    for y, x in zip(ys, xs):
        base = [cols[f][y, x] for f in range(frames)]
        assert (blend(base) == want[y, x, :3]).all()
        found = []
        for j in range(frames):
            for j2 in range(frames):
                if j2 != j:
                    alt = list(base)
                    alt[j] = cols[j2][y, x]
                    if (blend(alt) == got[y, x, :3]).all():
                        found.append("frame %d blended with frame %d's colour" % (j, j2))
            for k in (1, 2, 3):
                alt = list(base)
                alt[j] = part[(j, k)][y, x] * np.float32(0.25)
                if (blend(alt) == got[y, x, :3]).all():
                    found.append("frame %d holds only its first %d samples (x 1/4)" % (j, k))
            alt = list(base)
            alt[j] = np.zeros(3, np.float32)
            if (blend(alt) == got[y, x, :3]).all():
                found.append("frame %d's colour is zero" % j)
            alt = [c for f, c in enumerate(base) if f != j]
            # (a missing blend shifts nothing else: the lerp factor belongs to the frame number)
            acc = np.zeros(3, np.float32)
            for f, c in enumerate(base):
                if f == j:
                    continue
                lerp = np.float32(np.float32(f) / np.float32(f + 1))
                acc = acc * lerp + c * np.float32(np.float32(1) - lerp)
            if (acc == got[y, x, :3]).all():
                found.append("frame %d's blend is missing" % j)
        # the implied colour per frame, for the eye
        implied = []
        for j in range(frames):
            wj = np.float32(1.0) / np.float32(j + 1)
            for f in range(j + 1, frames):
                wj = wj * np.float32(np.float32(f) / np.float32(f + 1))
            implied.append(base[j] + (got[y, x, :3] - want[y, x, :3]) / wj)
        print("(x %3d, y %3d) got %s want %s -> %s" % (x, y, got[y, x, :3], want[y, x, :3], "; ".join(found) if found else "no simple hypothesis"))
        if not found:
            for j in range(frames):
                print("      if frame %d: colour %s instead of %s" % (j, implied[j], base[j]))


if __name__ == "__main__":
    main()
