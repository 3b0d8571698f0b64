"""Round 5, first GPU call: what exactly is wrong when test_sharded_batches_match_per_frame_exchange fails on the
-DTPT_TAIL_HELPERS=1 build after the other tests of tests/test_gpu_api.py (profiles/r04/r04_run24.log)?

Runs the predecessors' scenarios and the failing one in ONE process, `--reps` times, and on a mismatch says which pixels differ and
which hypothesis about the blend chain reproduces them (a frame's blend missing / applied twice / a stale colour plane), using the
oracle's per-frame colours and a numpy restatement of Test.cpp:293-295.

    TPT_LIB_DIR=tools/_variants/helpers python tools/helpers_diag.py --reps 6
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle_lib import FLAG_PROGRESSIVE, SEED_PER_PIXEL, Oracle  # noqa: E402
from toypathtracer_amd import api as tpt  # noqa: E402


def frame_colours(o, w, h, spp, frames):
    """per-frame colour (what the trace kernel writes) from the oracle: a frame rendered without kFlagProgressive on a zeroed buffer"""
    s, m = o.default_scene()
    cam = o.default_camera(w, h)
    cols = []
    for f in range(frames):
        _, bb = o.render(s, m, cam, w, h, spp, f, 0, seed_mode=SEED_PER_PIXEL)
        cols.append(bb[..., :3].copy())
    return cols


def blend(seq, cols, h, w):
    """the progressive blends of frames `seq` (frame numbers; the lerp factor belongs to the frame number, Test.cpp:272-276)"""
    acc = np.zeros((h, w, 3), np.float32)
    for f in seq:
        lerp = np.float32(np.float32(f) / np.float32(f + 1))
        acc = acc * lerp + cols[f] * np.float32(np.float32(1) - lerp)
    return acc


def scenario_single_rank():
    tpt.comm_init(tpt.comm_get_unique_id(), 1, 0, 8)
    try:
        for (w, h, frames) in [(256, 144, 5), (160, 90, 3)]:
            img = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
            for f in range(frames):
                tpt.UpdateTest(0.0, f, w, h, FLAG_PROGRESSIVE)
                tpt.draw_sharded(0.0, f, w, h, img.data_ptr(), FLAG_PROGRESSIVE)
            tpt.sharded_finish()
    finally:
        tpt.comm_destroy()


def scenario_loopback(n):
    w, h, frames, stripe = 200, 120, 20, 8
    tpt.comm_init_loopback(n, stripe)
    try:
        img = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
        for f in range(frames):
            tpt.UpdateTest(0.0, f, w, h, FLAG_PROGRESSIVE)
            tpt.draw_sharded(0.0, f, w, h, img.data_ptr(), FLAG_PROGRESSIVE)
        tpt.sharded_finish()
    finally:
        tpt.comm_destroy()


def scenario_stream(w=1280, h=720, frames=24):
    """a plain stream of C2 frames and a blocking wait: what makes the helpers launch at all"""
    tile = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    for f in range(frames):
        tpt.UpdateTest(0.0, f, w, h, FLAG_PROGRESSIVE)
        tpt.draw_device(0.0, f, w, h, tile.data_ptr(), FLAG_PROGRESSIVE)
    tpt.synchronize()


def failing(o, cols, want, batches=(3, 1, 4)):
    w, h, stripe, n = 200, 120, 8, 4
    tpt.comm_init_loopback(n, stripe)
    try:
        img = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
        f = 0
        for k in batches:
            tpt.UpdateTest(0.0, f, w, h, FLAG_PROGRESSIVE)
            tpt.draw_sharded_batch(0.0, f, k, w, h, img.data_ptr(), FLAG_PROGRESSIVE)
            f += k
        tpt.sharded_finish()
        got = img.cpu().numpy()
    finally:
        tpt.comm_destroy()
    mine = (np.arange(h) // stripe) % n == 0
    ok = got[mine].tobytes() == want[mine].tobytes() and not got[~mine].any()
    if ok:
        return True
    bad = (got[..., :3] != want[..., :3]).any(axis=2)
    rows = np.nonzero(bad.any(axis=1))[0]
    print("  MISMATCH: %d pixels in %d rows (rows %s); of them rank 0's: %d; non-zero pixels outside rank 0's rows: %d" % (
        bad.sum(), len(rows), rows[:12].tolist(), bad[mine].sum(), int(got[~mine].any(axis=2).sum())))
    ys, xs = np.nonzero(bad)
    for y, x in list(zip(ys, xs))[:4]:
        print("    (%d, %d): got %s want %s" % (x, y, got[y, x, :3], want[y, x, :3]))
    # hypotheses on the differing pixels of rank 0's rows
    total = sum(batches)
    hyp = {}
    for j in range(total):
        hyp["frame %d's blend missing" % j] = [f for f in range(total) if f != j]
        hyp["frame %d blended twice" % j] = [f for f in range(total) for _ in range(2 if f == j else 1)]
    hyp["only the first batch"] = list(range(batches[0]))
    hyp["without the last batch"] = list(range(total - batches[-1]))
    sel = bad & mine[:, None]
    for name, seq in hyp.items():
        alt = blend(seq, cols, h, w)
        if sel.any() and (alt[sel] == got[..., :3][sel]).all():
            print("    -> reproduced by: %s" % name)
    for j in range(total):  # a pixel that carries another frame's colour in place of its own
        for j2 in range(total):
            if j2 == j:
                continue
            acc = np.zeros((h, w, 3), np.float32)
            for f in range(total):
                lerp = np.float32(np.float32(f) / np.float32(f + 1))
                acc = acc * lerp + cols[j2 if f == j else f] * np.float32(np.float32(1) - lerp)
            if sel.any() and (acc[sel] == got[..., :3][sel]).all():
                print("    -> reproduced by: frame %d blended with the colour plane of frame %d" % (j, j2))
    return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--order", default="single,loop2,loop4,stream,fail", help="comma list of single, loop2, loop4, stream, fail")
    args = ap.parse_args()
    o = Oracle.get()
    w, h = 200, 120
    cols = frame_colours(o, w, h, 4, 8)
    _, bo = o.render_frames(w, h, 4, 8, seed_mode=SEED_PER_PIXEL)
    want = np.frombuffer(bo.tobytes(), np.float32).reshape(h, w, 4)
    assert (blend(range(8), cols, h, w) == want[..., :3]).all(), "the numpy restatement of the blend chain differs from the oracle"
    tpt.InitializeTest()
    print("library:", tpt.library_path(), " TPT_TAIL_HELPERS =", os.environ.get("TPT_TAIL_HELPERS"))
    fails = 0
    for rep in range(args.reps):
        res = []
        for what in args.order.split(","):
            if what == "single":
                scenario_single_rank()
            elif what == "loop2":
                scenario_loopback(2)
            elif what == "loop4":
                scenario_loopback(4)
            elif what == "stream":
                scenario_stream()
            elif what == "fail":
                ok = failing(o, cols, want)
                res.append(ok)
                fails += 0 if ok else 1
        print("rep %d: %s" % (rep, ["ok" if r else "FAIL" for r in res]), flush=True)
    print("failures: %d of %d" % (fails, args.reps))
    tpt.ShutdownTest()


if __name__ == "__main__":
    main()
