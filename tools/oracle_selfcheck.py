"""Round 6: is the checker deterministic on the GPU box's host?    The case of the two round-5 test
failures -- 200x120, 4 spp, 8 progressive frames, per-pixel seeds -- rendered over and over
  (1) by oracle/_build/oracle_soak: the oracle as the checker is built, without -mfma, without OpenMP, under ThreadSanitizer;
  (2) in THIS process the way the GPU suite uses it (ctypes, torch imported, the HIP library initialised and rendering beside it);
  (3) by the reference-compiled per-pixel build (oracle/_ref/libtpt_ref_perpixel.so, enkiTS threads) when it travelled.
Every result is compared with the hash computed in the build container (EXPECT).   python tools/oracle_selfcheck.py [reps]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["TPT_ORACLE_REDUNDANT"] = "0"
EXPECT, RAYS = "e63fd48d", 3556727
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = os.path.join(ROOT, "oracle", "_build")


def sh(cmd, t=120):
    t0 = time.time()
    try:
        out = subprocess.run(cmd, shell=True, capture_output=True, text=True, timeout=t)
        txt = (out.stdout + out.stderr).strip().splitlines()
    except subprocess.TimeoutExpired:
        txt = ["TIMEOUT after %d s" % t]
    print("$ %s   [%.1f s]" % (cmd, time.time() - t0))
    for ln in txt[-6:]:
        print("    " + ln[:300])
    sys.stdout.flush()


print("host: %d cpus; %s" % (os.cpu_count(), subprocess.getoutput("grep -m1 'model name' /proc/cpuinfo; grep -m1 microcode /proc/cpuinfo").replace("\n", "; ")))
def soaks():
    sh("%s/oracle_soak %d 0 %s" % (B, max(20, reps // 4), EXPECT), 200)      # every hardware thread (256 on the GPU boxes: ~0.5 s per render, spin-waiting OpenMP threads)
    sh("%s/oracle_soak %d 24 %s" % (B, reps, EXPECT), 200)
    sh("%s/oracle_soak_nofma %d 64 %s" % (B, reps, EXPECT), 200)
    sh("for i in $(seq 16); do %s/oracle_soak_plain %d 1 %s | tail -1 & done; wait" % (B, max(4, reps // 16), EXPECT), 200)
    # (ThreadSanitizer refuses the address-space layout of some kernels: retry without ASLR)
    sh("%s/oracle_soak_tsan 3 64 %s 2>&1 | tail -4 | grep -q 'differ from the first' && %s/oracle_soak_tsan 3 64 %s 2>&1 | tail -3 || setarch x86_64 -R %s/oracle_soak_tsan 3 64 %s 2>&1 | tail -4" % (B, EXPECT, B, EXPECT, B, EXPECT), 300)


import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle_lib import SEED_PER_PIXEL, Oracle, Ref, fnv1a  # noqa: E402
from toypathtracer_amd import api as tpt  # noqa: E402

gpu = torch.cuda.is_available()
if gpu:
    tpt.InitializeTest()
    tile = torch.zeros((720, 1280, 4), dtype=torch.float32, device="cuda")
o = Oracle.get()
bad = {}
t0 = time.time()
for r in range(reps):
    if gpu:  # the GPU busy beside the oracle, as in the suite
        for f in range(4):
            tpt.UpdateTest(0.0, f, 1280, 720, 2)
            tpt.draw_device(0.0, f, 1280, 720, tile.data_ptr(), 2)
    rays, bb = o.render_frames(200, 120, 4, 8, seed_mode=SEED_PER_PIXEL, threads=0 if r % 2 == 0 else 24)
    key = ("%08x" % fnv1a(bb), rays)
    if key != (EXPECT, RAYS):
        bad[key] = bad.get(key, 0) + 1
if gpu:
    tpt.synchronize()
print("in-process oracle (ctypes, torch imported, GPU %s): %d of %d results differ from %s: %s   [%.1f s]" % ("busy" if gpu else "absent", sum(bad.values()), reps, EXPECT, bad, time.time() - t0))
if Ref.available("perpixel"):
    ref = Ref.get("perpixel")
    badr = {}
    t0 = time.time()
    for r in range(reps):
        rays, bb = ref.render_frames(200, 120, 4, 8)
        key = ("%08x" % fnv1a(bb), rays)
        if key != (EXPECT, RAYS):
            badr[key] = badr.get(key, 0) + 1
    print("reference-compiled per-pixel build (enkiTS): %d of %d results differ from %s: %s   [%.1f s]" % (sum(badr.values()), reps, EXPECT, badr, time.time() - t0))
if gpu:
    tpt.ShutdownTest()
soaks()
