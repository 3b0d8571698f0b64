#!/usr/bin/env python3
"""Synchronous DrawTest(host float* backbuffer) per frame at C2 (1280x720x4spp), the reference's own calling contract, by
look-ahead depth (tptSetHostLookahead), workgroups per launch (env TPT_GRID_FILL), host copy threads (tptSetHostCopyThreads), with
and without the trusted-buffer mode (no upload).  bench.py's `drawtest_host_ms` is the all-defaults line of this table.  Usage: python tools/host_drawtest_rate.py [w h]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from toypathtracer_amd import api  # noqa: E402

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1280, 720)
for fill in os.environ.get("FILLS", ",100").split(","):
    if fill:
        os.environ["TPT_GRID_FILL"] = fill
    api.InitializeTest()
    api.set_samples_per_pixel(4)
    api.set_host_copy_threads(4)
    for trusted in (False, True):
        api.set_host_buffer_mode(trusted)
        for ahead in [int(x) for x in os.environ.get("AHEADS", "2,3,4,6,8,12").split(",")]:
            bench.HOST_LOOKAHEAD = ahead
            api.set_host_lookahead(ahead)
            best = min(bench.drawtest_host_path(api, w, h, frames=48) for _ in range(3))
            print("grid fill %-7s trusted %d  look-ahead %2d: %.3f ms per frame, %.0f Mray/s" % (fill or "default", trusted, ahead, best[0], best[1]), flush=True)
    bench.HOST_LOOKAHEAD = 2
    api.set_host_lookahead(2)
    api.set_host_buffer_mode(False)
    for threads in (1, 2, 4, 8):
        api.set_host_copy_threads(threads)
        best = min(bench.drawtest_host_path(api, w, h, frames=48) for _ in range(3))
        print("grid fill %-7s look-ahead 2, copy threads %d: %.3f ms per frame, %.0f Mray/s" % (fill or "default", threads, best[0], best[1]), flush=True)
    api.ShutdownTest()
